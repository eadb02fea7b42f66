// Scene assembly (see Scene.h): OBJ scenes as MainFrm.cpp:228-492 builds them, a
// procedural atrium standing in for the absent Sponza meshes, instancing, PNG decoding.
#include "Scene.h"
#include "ObjLoader.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <zlib.h>

namespace pth {

// ---------------------------------------------------------------------------------
void HostScene::Finalize(bool sortMeshes)
{
	if (sortMeshes) {
		std::stable_sort(meshes.begin(), meshes.end(), [](const HostMesh& a, const HostMesh& b) {
			return a.vertices.size() > b.vertices.size();
		});
	}
	mVertices.clear(); mMeshDesc.clear(); mMatDesc.clear();
	for (auto& m : materials) {
		mipt_material d = {};
		if (!m.tex.empty() && m.width > 0 && m.height > 0) {
			d.tex_data = m.tex.data(); d.tex_width = m.width; d.tex_height = m.height; d.tex_format = MIPT_TEX_BGRA8;
		}
		if (!m.nml.empty() && m.nmlWidth > 0 && m.nmlHeight > 0) {
			d.nml_data = m.nml.data(); d.nml_width = m.nmlWidth; d.nml_height = m.nmlHeight; d.nml_format = MIPT_TEX_BGRA8;
		}
		d.diffuse[0] = m.diffuse[0]; d.diffuse[1] = m.diffuse[1]; d.diffuse[2] = m.diffuse[2];
		mMatDesc.push_back(d);
	}
	size_t total = 0;
	for (auto& m : meshes) total += m.vertices.size();
	mVertices.reserve(total);
	for (auto& m : meshes) {
		m.materialIndex = -1;
		for (size_t i = 0; i < materials.size(); i++)      // first name match wins (MainFrm.cpp:445-457)
			if (materials[i].name == m.material) { m.materialIndex = (int32_t)i; break; }
		mipt_mesh d = {};
		d.first_triangle = (int64_t)(mVertices.size() / 3);
		d.num_triangles = (int64_t)(m.vertices.size() / 3);
		d.material = m.materialIndex;
		d.is_light = m.light ? 1 : 0;
		mMeshDesc.push_back(d);
		mVertices.insert(mVertices.end(), m.vertices.begin(), m.vertices.end());
	}
	mDesc.vertices = mVertices.data();
	mDesc.num_triangles = (int64_t)(mVertices.size() / 3);
	mDesc.meshes = mMeshDesc.data();
	mDesc.num_meshes = (int32_t)mMeshDesc.size();
	mDesc.materials = mMatDesc.data();
	mDesc.num_materials = (int32_t)mMatDesc.size();
}

// ---------------------------------------------------------------------------------
HostScene* AssembleObjScene(const char* path, bool flipWinding)
{
	ObjMesh obj;
	if (!LoadObj(path, &obj)) return nullptr;
	std::string p(path);
	size_t slash = p.find_last_of("/\\");
	std::string dir = (slash == std::string::npos) ? std::string() : p.substr(0, slash + 1);

	HostScene* scene = new HostScene();
	scene->label = std::string("obj:") + (slash == std::string::npos ? p : p.substr(slash + 1));
	// materials, MainFrm.cpp:231-312: diffuse is stored BGR; a material that names a diffuse
	// map is kept only if that map loads, and its normal map is looked at only then
	for (auto& m : obj.materials) {
		HostMaterial t;
		t.name = m.name;
		t.diffuse[0] = m.Kd[2]; t.diffuse[1] = m.Kd[1]; t.diffuse[2] = m.Kd[0];
		if (!m.map_Kd.empty()) {
			if (!LoadImageFile((dir + m.map_Kd).c_str(), t.tex, t.width, t.height)) continue;
			if (!m.map_Bump.empty()) LoadImageFile((dir + m.map_Bump).c_str(), t.nml, t.nmlWidth, t.nmlHeight);
		}
		scene->materials.push_back(std::move(t));
	}
	// one mesh per usemtl group, MainFrm.cpp:315-486: first three corners of each face
	const int i2 = flipWinding ? 2 : 1, i3 = flipWinding ? 1 : 2;     // :378-383
	for (auto& g : obj.matGroups) {
		HostMesh mesh;
		mesh.material = g.name;
		mesh.vertices.reserve((size_t)g.numFaces * 3);
		for (unsigned h = g.firstFace; h < g.firstFace + g.numFaces; h++) {
			const ObjFace& f = obj.faces[h];
			const int corner[3] = { 0, i2, i3 };
			for (int c = 0; c < 3; c++) {
				mipt_vertex v = {};
				int vi = obj.faceVertices[(size_t)f.firstVertex + corner[c]];
				if (vi >= 0 && (size_t)vi * 3 + 2 < obj.vertices.size()) {
					v.pos[0] = obj.vertices[3 * vi]; v.pos[1] = obj.vertices[3 * vi + 1]; v.pos[2] = obj.vertices[3 * vi + 2];
				}
				if (f.firstNormal >= 0) {
					int ni = obj.faceNormals[(size_t)f.firstNormal + corner[c]];
					if (ni >= 0 && (size_t)ni * 3 + 2 < obj.normals.size()) {
						v.normal[0] = obj.normals[3 * ni]; v.normal[1] = obj.normals[3 * ni + 1]; v.normal[2] = obj.normals[3 * ni + 2];
					}
				}
				if (f.firstTexCoord >= 0) {                                   // :367
					int ti = obj.faceTexCoords[(size_t)f.firstTexCoord + corner[c]];
					if (ti >= 0 && (size_t)ti * 2 + 1 < obj.texCoords.size()) {
						v.tex[0] = obj.texCoords[2 * ti]; v.tex[1] = obj.texCoords[2 * ti + 1];
					}
				}
				mesh.vertices.push_back(v);
			}
		}
		scene->meshes.push_back(std::move(mesh));
	}
	scene->Finalize(true);
	return scene;
}

// ---------------------------------------------------------------------------------
// Procedural atrium
namespace {

struct V3 { float x, y, z; };
static inline V3 operator+(V3 a, V3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
static inline V3 operator-(V3 a, V3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
static inline V3 operator*(V3 a, float s) { return { a.x * s, a.y * s, a.z * s }; }
static inline V3 cross(V3 a, V3 b) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
static inline V3 norm(V3 a) { float l = std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); return l > 0 ? a * (1.0f / l) : V3{ 0, 1, 0 }; }

static inline uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
static inline float hash01(uint32_t a, uint32_t b, uint32_t c) { return (float)(hash32(a * 0x9e3779b1U ^ hash32(b + 0x85ebca6bU) ^ hash32(c * 0xc2b2ae35U + 7U)) >> 8) * (1.0f / 16777216.0f); }

// tileable value noise on a period x period lattice
static float valueNoise(float x, float y, int period, uint32_t seed)
{
	int xi = (int)std::floor(x), yi = (int)std::floor(y);
	float fx = x - xi, fy = y - yi;
	auto wrap = [period](int v) { v %= period; return v < 0 ? v + period : v; };
	int x0 = wrap(xi), x1 = wrap(xi + 1), y0 = wrap(yi), y1 = wrap(yi + 1);
	float sx = fx * fx * (3 - 2 * fx), sy = fy * fy * (3 - 2 * fy);
	float a = hash01((uint32_t)x0, (uint32_t)y0, seed), b = hash01((uint32_t)x1, (uint32_t)y0, seed);
	float c = hash01((uint32_t)x0, (uint32_t)y1, seed), d = hash01((uint32_t)x1, (uint32_t)y1, seed);
	return (a + (b - a) * sx) + ((c + (d - c) * sx) - (a + (b - a) * sx)) * sy;
}
static float fbm(float u, float v, uint32_t seed)
{
	float s = 0, amp = 0.5f; int per = 8;
	for (int o = 0; o < 4; o++) { s += amp * valueNoise(u * per, v * per, per, seed + (uint32_t)o); amp *= 0.5f; per *= 2; }
	return s;
}

struct Builder {
	HostScene* scene = nullptr;     // null = count only
	std::vector<int64_t> counts;
	int64_t total = 0;
	float scale = 1.0f;             // world scale applied to positions
	float uvScale = 1.0f / 400.0f;  // texture repeats per (unscaled) unit

	void tri(int mat, V3 a, V3 b, V3 c, V3 na, V3 nb, V3 nc, float ua, float va, float ub, float vb, float uc, float vc)
	{
		total++;
		if (!scene) return;
		HostMesh& m = scene->meshes[(size_t)mat];
		const V3 P[3] = { a, b, c }, N[3] = { na, nb, nc };
		const float U[3] = { ua, ub, uc }, V[3] = { va, vb, vc };
		for (int i = 0; i < 3; i++) {
			mipt_vertex v;
			v.pos[0] = P[i].x * scale; v.pos[1] = P[i].y * scale; v.pos[2] = P[i].z * scale;
			v.normal[0] = N[i].x; v.normal[1] = N[i].y; v.normal[2] = N[i].z;
			v.tex[0] = U[i]; v.tex[1] = V[i];
			m.vertices.push_back(v);
		}
	}
	// parametric surface patch: P(s,t), s,t in [0,1], tessellated nu x nv, normals from the patch
	template <class F>
	void patch(int mat, int nu, int nv, F P, bool flip, float u0, float v0, float du, float dv)
	{
		nu = std::max(nu, 1); nv = std::max(nv, 1);
		if (!scene) { total += (int64_t)2 * nu * nv; return; }
		std::vector<V3> pts((size_t)(nu + 1) * (nv + 1)), nrm(pts.size());
		for (int j = 0; j <= nv; j++)
			for (int i = 0; i <= nu; i++) pts[(size_t)j * (nu + 1) + i] = P((float)i / nu, (float)j / nv);
		const float e = 1e-3f;
		for (int j = 0; j <= nv; j++)
			for (int i = 0; i <= nu; i++) {
				float s = (float)i / nu, t = (float)j / nv;
				V3 ds = P(std::min(s + e, 1.0f), t) - P(std::max(s - e, 0.0f), t);
				V3 dt = P(s, std::min(t + e, 1.0f)) - P(s, std::max(t - e, 0.0f));
				V3 n = norm(cross(ds, dt));
				nrm[(size_t)j * (nu + 1) + i] = flip ? n * -1.0f : n;
			}
		for (int j = 0; j < nv; j++)
			for (int i = 0; i < nu; i++) {
				size_t a = (size_t)j * (nu + 1) + i, b = a + 1, c = a + (nu + 1), d = c + 1;
				float ua = u0 + du * i / nu, ub = u0 + du * (i + 1) / nu, va = v0 + dv * j / nv, vb = v0 + dv * (j + 1) / nv;
				if (!flip) {
					tri(mat, pts[a], pts[b], pts[d], nrm[a], nrm[b], nrm[d], ua, va, ub, va, ub, vb);
					tri(mat, pts[a], pts[d], pts[c], nrm[a], nrm[d], nrm[c], ua, va, ub, vb, ua, vb);
				} else {
					tri(mat, pts[a], pts[d], pts[b], nrm[a], nrm[d], nrm[b], ua, va, ub, vb, ub, va);
					tri(mat, pts[a], pts[c], pts[d], nrm[a], nrm[c], nrm[d], ua, va, ua, vb, ub, vb);
				}
			}
	}
	void quad(int mat, V3 o, V3 eu, V3 ev, int nu, int nv, bool flip = false)
	{
		float lu = std::sqrt(eu.x * eu.x + eu.y * eu.y + eu.z * eu.z), lv = std::sqrt(ev.x * ev.x + ev.y * ev.y + ev.z * ev.z);
		// texture coordinates follow world position so that they run outside [0,1] and below 0
		float u0 = (o.x + o.z * 0.37f) * uvScale, v0 = (o.y + o.z * 0.61f) * uvScale;
		patch(mat, nu, nv, [=](float s, float t) { return o + eu * s + ev * t; }, flip, u0, v0, lu * uvScale, lv * uvScale);
	}
	void box(int mat, V3 lo, V3 hi, int n)
	{
		V3 d = hi - lo;
		quad(mat, { lo.x, lo.y, hi.z }, { d.x, 0, 0 }, { 0, d.y, 0 }, n, n);           // +z
		quad(mat, { hi.x, lo.y, lo.z }, { -d.x, 0, 0 }, { 0, d.y, 0 }, n, n);          // -z
		quad(mat, { hi.x, lo.y, hi.z }, { 0, 0, -d.z }, { 0, d.y, 0 }, n, n);          // +x
		quad(mat, { lo.x, lo.y, lo.z }, { 0, 0, d.z }, { 0, d.y, 0 }, n, n);           // -x
		quad(mat, { lo.x, hi.y, hi.z }, { d.x, 0, 0 }, { 0, 0, -d.z }, n, n);          // +y
		quad(mat, { lo.x, lo.y, lo.z }, { d.x, 0, 0 }, { 0, 0, d.z }, n, n);           // -y
	}
	void cylinder(int mat, V3 base, float r0, float r1, float h, int segs, int rings, float flute)
	{
		const float twoPi = 6.2831853f;
		patch(mat, segs, rings, [=](float s, float t) {
			float a = twoPi * s;
			float r = (r0 + (r1 - r0) * t) * (1.0f + flute * std::cos(a * 12.0f));
			return V3{ base.x + r * std::cos(a), base.y + h * t, base.z - r * std::sin(a) };
		}, false, base.x * uvScale, base.y * uvScale, twoPi * r0 * uvScale, h * uvScale);
	}
	void sphere(int mat, V3 c, float r, int segs, int rings, float bump, uint32_t seed)
	{
		const float twoPi = 6.2831853f, pi = 3.14159265f;
		patch(mat, segs, rings, [=](float s, float t) {
			float a = twoPi * s, b = pi * (0.02f + 0.96f * t);
			float rr = r * (1.0f + bump * (fbm(s * 2, t, seed) - 0.5f));
			return V3{ c.x + rr * std::sin(b) * std::cos(a), c.y - rr * std::cos(b), c.z - rr * std::sin(b) * std::sin(a) };
		}, false, c.x * uvScale, c.y * uvScale, 2.0f, 1.0f);
	}
	// semicircular arch between x0 and x1 at height y0 in the plane z, thickness th (along z), band width bw
	void arch(int mat, float x0, float x1, float y0, float z, float th, float bw, int segs)
	{
		const float pi = 3.14159265f;
		float cx = 0.5f * (x0 + x1), R = 0.5f * (x1 - x0), r = R - bw;
		for (int side = 0; side < 2; side++) {                       // the two faces of the band
			float zz = side ? z + 0.5f * th : z - 0.5f * th;
			patch(mat, segs, 1, [=](float s, float t) {
				float a = pi * s, rr = r + (R - r) * t;
				return V3{ cx - rr * std::cos(a), y0 + rr * std::sin(a), zz };
			}, side == 0, cx * uvScale, y0 * uvScale, pi * R * uvScale, bw * uvScale);
		}
		patch(mat, segs, 1, [=](float s, float t) {                   // intrados
			float a = pi * s;
			return V3{ cx - r * std::cos(a), y0 + r * std::sin(a), z - 0.5f * th + th * t };
		}, false, cx * uvScale, z * uvScale, pi * r * uvScale, th * uvScale);
	}
	void curtain(int mat, float x0, float x1, float ytop, float ybot, float z, int nu, int nv, uint32_t seed, bool flip)
	{
		float ph = 6.2831853f * hash01(seed, 3, 9);
		patch(mat, nu, nv, [=](float s, float t) {
			float w = 28.0f * std::sin(s * 25.0f + ph) * (0.3f + 0.7f * t) + 10.0f * std::sin(t * 9.0f + s * 4.0f);
			return V3{ x0 + (x1 - x0) * s, ytop + (ybot - ytop) * t, z + w };
		}, flip, 0.0f, 0.0f, 2.0f, 2.0f);
	}
};

struct MatDef { const char* name; uint8_t b, g, r; int pattern; bool textured; };
static const MatDef kMats[] = {
	{ "floor", 150, 170, 190, 1, true },       { "bricks", 120, 140, 180, 2, true },   { "arch", 170, 185, 200, 0, true },
	{ "column_a", 180, 190, 205, 0, true },    { "column_b", 165, 180, 200, 0, true }, { "column_c", 150, 165, 185, 0, true },
	{ "ceiling", 140, 160, 175, 1, true },     { "roof", 90, 110, 160, 2, true },      { "details", 130, 150, 170, 0, true },
	{ "fabric_a", 60, 60, 200, 3, true },      { "fabric_c", 200, 90, 60, 3, true },   { "fabric_d", 200, 80, 70, 3, true },
	{ "fabric_e", 70, 180, 70, 3, true },      { "fabric_f", 80, 190, 90, 3, true },   { "fabric_g", 210, 100, 80, 3, true },
	{ "vase", 120, 150, 190, 0, true },        { "vase_round", 110, 140, 150, 0, true }, { "vase_hanging", 100, 120, 130, 0, true },
	{ "Material__25", 140, 170, 200, 0, true },  /* lion */
	{ "Material__042", 160, 170, 180, 2, true }, /* background */
	{ "flagpole", 90, 100, 110, 0, true },     { "chain", 80, 80, 80, 0, true },       { "leaf", 60, 150, 70, 0, true },
	{ "VasePlant", 70, 140, 80, 0, true },     { "Material__47", 150, 150, 150, 0, false },
};
static const int kNumMats = (int)(sizeof(kMats) / sizeof(kMats[0]));
enum { M_FLOOR, M_BRICKS, M_ARCH, M_COL_A, M_COL_B, M_COL_C, M_CEIL, M_ROOF, M_DETAILS, M_FAB_A, M_FAB_C, M_FAB_D, M_FAB_E,
       M_FAB_F, M_FAB_G, M_VASE, M_VASE_ROUND, M_VASE_HANG, M_LION, M_BACKGROUND, M_FLAGPOLE, M_CHAIN, M_LEAF, M_PLANT, M_PLAIN };

static void makeTexture(HostMaterial& m, const MatDef& d, int size, uint32_t seed)
{
	m.width = m.height = m.nmlWidth = m.nmlHeight = size;
	m.tex.resize((size_t)size * size); m.nml.resize((size_t)size * size);
	std::vector<float> hgt((size_t)size * size);
	for (int y = 0; y < size; y++)
		for (int x = 0; x < size; x++) {
			float u = (x + 0.5f) / size, v = (y + 0.5f) / size;
			float n = fbm(u, v, seed);
			float k = 0.75f + 0.5f * (n - 0.5f), h = n;
			if (d.pattern == 1) {                 // tiles
				float fu = u * 8 - std::floor(u * 8), fv = v * 8 - std::floor(v * 8);
				bool joint = fu < 0.04f || fv < 0.04f;
				k *= joint ? 0.55f : (((int)(u * 8) + (int)(v * 8)) & 1 ? 1.0f : 0.85f);
				h = joint ? 0.0f : 0.6f + 0.4f * n;
			} else if (d.pattern == 2) {          // bricks
				float row = std::floor(v * 16), fu = u * 8 + ((int)row & 1 ? 0.5f : 0.0f);
				fu -= std::floor(fu); float fv = v * 16 - row;
				bool joint = fu < 0.05f || fv < 0.1f;
				k *= joint ? 0.5f : 0.8f + 0.4f * hash01((uint32_t)row, (uint32_t)std::floor(u * 8 + ((int)row & 1 ? 0.5f : 0.0f)), seed);
				h = joint ? 0.0f : 0.7f + 0.3f * n;
			} else if (d.pattern == 3) {          // woven stripes
				float s = 0.5f + 0.5f * std::sin(u * 6.2831853f * 24);
				k *= 0.7f + 0.3f * s; h = 0.5f * s + 0.5f * n;
			}
			auto ch = [k](uint8_t c) { float f = c * k; return (uint32_t)(f < 0 ? 0 : f > 255 ? 255 : f); };
			m.tex[(size_t)y * size + x] = ch(d.b) | (ch(d.g) << 8) | (ch(d.r) << 16) | (255u << 24);
			hgt[(size_t)y * size + x] = h;
		}
	for (int y = 0; y < size; y++)
		for (int x = 0; x < size; x++) {
			float hx = hgt[(size_t)y * size + (x + 1) % size] - hgt[(size_t)y * size + (x + size - 1) % size];
			float hy = hgt[(size_t)((y + 1) % size) * size + x] - hgt[(size_t)((y + size - 1) % size) * size + x];
			V3 n = norm({ -hx * 2.0f, -hy * 2.0f, 1.0f });
			auto enc = [](float f) { float q = 255.0f * (0.5f + 0.5f * f); return (uint32_t)(q < 0 ? 0 : q > 255 ? 255 : q); };
			// red = tangent x, green = tangent y, blue = along the surface normal
			m.nml[(size_t)y * size + x] = enc(n.z) | (enc(n.y) << 8) | (enc(n.x) << 16) | (255u << 24);
		}
}

static void generate(Builder& B, float T, uint32_t seed)
{
	auto seg = [T](float base, int lo) { return std::max(lo, (int)std::lround(base * T)); };
	// floor of court and aisles, gallery floors, roofs
	B.quad(M_FLOOR, { -1700, 0, 1000 }, { 3400, 0, 0 }, { 0, 0, -2000 }, seg(34, 2), seg(20, 2));
	for (int s = -1; s <= 1; s += 2) {
		float zin = 450.0f * s, zout = 1000.0f * s;
		// aisle ceiling (faces down), gallery floor (faces up), aisle roof underside
		B.quad(M_CEIL, { -1700, 540, std::min(zin, zout) }, { 3400, 0, 0 }, { 0, 0, 550 }, seg(34, 2), seg(6, 1));
		B.quad(M_FLOOR, { -1700, 560, std::max(zin, zout) }, { 3400, 0, 0 }, { 0, 0, -550 }, seg(34, 2), seg(6, 1));
		B.quad(M_ROOF, { -1700, 1150, std::min(zin, zout) }, { 3400, 0, 0 }, { 0, 0, 550 }, seg(34, 2), seg(6, 1));
		// gallery edge (vertical band facing the court)
		if (s > 0) B.quad(M_DETAILS, { 1700, 540, zin }, { -3400, 0, 0 }, { 0, 20, 0 }, seg(34, 2), 1);
		else       B.quad(M_DETAILS, { -1700, 540, zin }, { 3400, 0, 0 }, { 0, 20, 0 }, seg(34, 2), 1);
		// outer wall, inside face
		if (s > 0) B.quad(M_BRICKS, { 1700, 0, zout }, { -3400, 0, 0 }, { 0, 1150, 0 }, seg(34, 2), seg(12, 2));
		else       B.quad(M_BRICKS, { -1700, 0, zout }, { 3400, 0, 0 }, { 0, 1150, 0 }, seg(34, 2), seg(12, 2));
		// cornice on top of the court side
		B.box(M_DETAILS, { -1700, 1150, std::min(zin, zin + 40 * s) }, { 1700, 1190, std::max(zin, zin + 40 * s) }, seg(3, 1));
	}
	// end walls
	B.quad(M_BACKGROUND, { 1700, 0, -1000 }, { 0, 0, 2000 }, { 0, 1150, 0 }, seg(20, 2), seg(12, 2));
	B.quad(M_BACKGROUND, { -1700, 0, 1000 }, { 0, 0, -2000 }, { 0, 1150, 0 }, seg(20, 2), seg(12, 2));
	// colonnades: two rows, two storeys
	const int ncol = 10;
	for (int s = -1; s <= 1; s += 2) {
		float z = 450.0f * s;
		for (int i = 0; i < ncol; i++) {
			float x = -1530.0f + 340.0f * i;
			int mcol = (i % 3 == 0) ? M_COL_A : (i % 3 == 1) ? M_COL_B : M_COL_C;
			B.box(M_DETAILS, { x - 55, 0, z - 55 }, { x + 55, 40, z + 55 }, seg(1.5f, 1));
			B.cylinder(mcol, { x, 40, z }, 42, 36, 340, seg(20, 6), seg(10, 1), 0.03f);
			B.box(M_DETAILS, { x - 52, 380, z - 52 }, { x + 52, 420, z + 52 }, seg(1.5f, 1));
			B.box(M_DETAILS, { x - 45, 560, z - 45 }, { x + 45, 590, z + 45 }, seg(1.5f, 1));
			B.cylinder(mcol, { x, 590, z }, 34, 30, 330, seg(18, 6), seg(9, 1), 0.0f);
			B.box(M_DETAILS, { x - 45, 920, z - 45 }, { x + 45, 950, z + 45 }, seg(1.5f, 1));
			if (i + 1 < ncol) {
				B.arch(M_ARCH, x + 40, x + 300, 400, z, 80, 45, seg(16, 4));
				B.arch(M_ARCH, x + 40, x + 300, 950, z, 70, 40, seg(14, 4));
				// wall above the arches up to the gallery / cornice
				if (s > 0) {
					B.quad(M_BRICKS, { x + 340, 530, z - 40 }, { -340, 0, 0 }, { 0, 10, 0 }, seg(4, 1), 1);
				} else {
					B.quad(M_BRICKS, { x, 530, z + 40 }, { 340, 0, 0 }, { 0, 10, 0 }, seg(4, 1), 1);
				}
				// curtains hang in every other upper bay
				if ((i & 1) == (s > 0 ? 0 : 1)) {
					int mf = M_FAB_A + ((i / 2 + (s > 0 ? 0 : 3)) % 6);
					B.curtain(mf, x + 50, x + 290, 930, 610, z - 25.0f * s, seg(24, 4), seg(20, 4), seed + (uint32_t)(i * 2 + (s > 0)), s < 0);
				}
			}
			// vases at every other column foot, hanging vases under the arches
			if (i % 2 == 0) {
				B.sphere(M_VASE_ROUND, { x, 95, z - 120.0f * s }, 55, seg(16, 6), seg(10, 4), 0.0f, seed);
				B.sphere(M_PLANT, { x, 190, z - 120.0f * s }, 60, seg(14, 6), seg(10, 4), 0.6f, seed + 17u * (uint32_t)i);
			} else {
				B.cylinder(M_CHAIN, { x + 170, 330, z }, 2.5f, 2.5f, 190, 6, seg(4, 1), 0.0f);
				B.sphere(M_VASE_HANG, { x + 170, 300, z }, 38, seg(12, 6), seg(8, 4), 0.0f, seed);
			}
		}
	}
	// large vases in the court, lion relief on the far wall, flag poles, an untextured plinth
	for (int i = 0; i < 4; i++) {
		float x = -1100.0f + 730.0f * i;
		B.sphere(M_VASE, { x, 110, (i & 1) ? 220.0f : -220.0f }, 105, seg(22, 8), seg(14, 6), 0.05f, seed + 5u);
		B.sphere(M_LEAF, { x, 290, (i & 1) ? 220.0f : -220.0f }, 120, seg(20, 8), seg(14, 6), 0.9f, seed + 31u * (uint32_t)(i + 1));
	}
	B.sphere(M_LION, { 1640, 330, 0 }, 170, seg(40, 8), seg(28, 6), 0.35f, seed + 99u);
	B.sphere(M_LION, { -1640, 330, 0 }, 170, seg(40, 8), seg(28, 6), 0.35f, seed + 199u);
	for (int i = 0; i < 6; i++) {
		float x = -1360.0f + 544.0f * i;
		B.box(M_FLAGPOLE, { x - 4, 700, -440 }, { x + 4, 708, -120 }, seg(2, 1));
		B.box(M_FLAGPOLE, { x - 4, 700, 120 }, { x + 4, 708, 440 }, seg(2, 1));
	}
	B.box(M_PLAIN, { -150, 0, -90 }, { 150, 60, 90 }, seg(3, 1));
}

} // namespace

HostScene* BuildAtrium(int sceneKind, int64_t targetTriangles, int textureSize, uint32_t seed)
{
	if (targetTriangles < 1000) targetTriangles = 1000;
	// tessellation factor by bisection on a counting run
	float lo = 0.02f, hi = 400.0f;
	for (int it = 0; it < 40; it++) {
		float mid = std::sqrt(lo * hi);
		Builder c; generate(c, mid, seed);
		if (c.total < targetTriangles) lo = mid; else hi = mid;
	}
	HostScene* scene = new HostScene();
	Builder B; B.scene = scene;
	// the generator works in Crytek-Sponza units (court 3400 x 900, camera preset (-1280,138,-50));
	// the default-Sponza preset (10,5,0) lives in a scene 1/120 of that size
	B.scale = (sceneKind == MIPT_SCENE_SPONZA) ? (1.0f / 120.0f) : 1.0f;
	scene->meshes.resize((size_t)kNumMats);
	scene->materials.resize((size_t)kNumMats);
	for (int i = 0; i < kNumMats; i++) {
		scene->meshes[(size_t)i].material = kMats[i].name;
		HostMaterial& m = scene->materials[(size_t)i];
		m.name = kMats[i].name;
		m.diffuse[0] = kMats[i].b / 255.0f; m.diffuse[1] = kMats[i].g / 255.0f; m.diffuse[2] = kMats[i].r / 255.0f;
		if (textureSize > 0 && kMats[i].textured) makeTexture(m, kMats[i], textureSize, seed * 131u + (uint32_t)i);
	}
	generate(B, hi, seed);
	// drop material groups that received no geometry
	std::vector<HostMesh> kept;
	for (auto& m : scene->meshes) if (!m.vertices.empty()) kept.push_back(std::move(m));
	scene->meshes.swap(kept);
	scene->Finalize(true);
	char buf[160];
	snprintf(buf, sizeof(buf), "synthetic-atrium(kind=%d,tris=%lld,tex=%d,seed=%u)", sceneKind, (long long)scene->NumTriangles(), textureSize, seed);
	scene->label = buf;
	return scene;
}

HostScene* BuildInstanced(const HostScene& base, int copies, float spacing)
{
	HostScene* scene = new HostScene();
	scene->materials = base.materials;
	scene->meshes.resize(base.meshes.size());
	int side = (int)std::ceil(std::sqrt((double)std::max(copies, 1)));
	for (size_t m = 0; m < base.meshes.size(); m++) {
		HostMesh& dst = scene->meshes[m];
		const HostMesh& src = base.meshes[m];
		dst.material = src.material; dst.light = src.light;
		dst.vertices.reserve(src.vertices.size() * (size_t)copies);
		for (int c = 0; c < copies; c++) {
			float dx = spacing * (float)(c % side), dz = spacing * (float)(c / side);
			for (mipt_vertex v : src.vertices) { v.pos[0] += dx; v.pos[2] += dz; dst.vertices.push_back(v); }
		}
	}
	scene->Finalize(true);
	char buf[200];
	snprintf(buf, sizeof(buf), "%s x%d instances", base.label.c_str(), copies);
	scene->label = buf;
	return scene;
}

// ---------------------------------------------------------------------------------
// PNG encoding on top of zlib: IHDR, one IDAT, IEND
static void put32(std::vector<uint8_t>& v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }
static void put_chunk(std::vector<uint8_t>& file, const char* type, const std::vector<uint8_t>& data)
{
	put32(file, (uint32_t)data.size());
	size_t start = file.size();
	file.insert(file.end(), type, type + 4);
	file.insert(file.end(), data.begin(), data.end());
	put32(file, (uint32_t)crc32(0, &file[start], (uInt)(file.size() - start)));
}

bool WritePng(const char* path, const std::vector<uint32_t>& bgra, int32_t width, int32_t height)
{
	if (width <= 0 || height <= 0 || bgra.size() < (size_t)width * height) return false;
	std::vector<uint8_t> raw(((size_t)width * 3 + 1) * height);
	for (int32_t y = 0; y < height; y++) {
		uint8_t* row = &raw[((size_t)width * 3 + 1) * y];
		row[0] = 0;                                   // filter type: none
		for (int32_t x = 0; x < width; x++) {
			uint32_t t = bgra[(size_t)y * width + x];
			row[1 + 3 * x] = (uint8_t)(t >> 16); row[2 + 3 * x] = (uint8_t)(t >> 8); row[3 + 3 * x] = (uint8_t)t;
		}
	}
	uLongf clen = compressBound((uLong)raw.size());
	std::vector<uint8_t> comp(clen);
	if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return false;
	comp.resize(clen);
	std::vector<uint8_t> file = { 137, 80, 78, 71, 13, 10, 26, 10 };
	std::vector<uint8_t> ihdr;
	put32(ihdr, (uint32_t)width); put32(ihdr, (uint32_t)height);
	ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);   // 8-bit RGB, no interlace
	put_chunk(file, "IHDR", ihdr);
	put_chunk(file, "IDAT", comp);
	put_chunk(file, "IEND", {});
	FILE* f = fopen(path, "wb");
	if (!f) return false;
	bool ok = fwrite(file.data(), 1, file.size(), f) == file.size();
	fclose(f);
	return ok;
}

// PNG decoding on top of zlib
static uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

static bool LoadPngUnguarded(const char* path, std::vector<uint32_t>& out, int32_t& width, int32_t& height)
{
	FILE* f = fopen(path, "rb");
	if (!f) return false;
	std::vector<uint8_t> file;
	uint8_t buf[65536]; size_t n;
	while ((n = fread(buf, 1, sizeof(buf), f)) > 0) file.insert(file.end(), buf, buf + n);
	fclose(f);
	static const uint8_t sig[8] = { 137, 80, 78, 71, 13, 10, 26, 10 };
	if (file.size() < 33 || memcmp(file.data(), sig, 8) != 0) return false;
	uint32_t w = 0, h = 0; int depth = 0, ctype = 0, interlace = 0;
	std::vector<uint8_t> idat; uint8_t pal[256][3] = {};
	size_t pos = 8;
	while (pos + 12 <= file.size()) {
		uint32_t len = be32(&file[pos]);
		const uint8_t* type = &file[pos + 4];
		const uint8_t* data = &file[pos + 8];
		if (pos + 12 + (size_t)len > file.size()) return false;
		if (!memcmp(type, "IHDR", 4)) {
			if (len != 13) return false;
			w = be32(data); h = be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12];
		} else if (!memcmp(type, "PLTE", 4)) {
			for (uint32_t i = 0; i < len / 3 && i < 256; i++) { pal[i][0] = data[3 * i]; pal[i][1] = data[3 * i + 1]; pal[i][2] = data[3 * i + 2]; }
		} else if (!memcmp(type, "IDAT", 4)) {
			idat.insert(idat.end(), data, data + len);
		} else if (!memcmp(type, "IEND", 4)) break;
		pos += 12 + (size_t)len;
	}
	if (!w || !h || interlace || (uint64_t)w * h > ((uint64_t)1 << 28)) return false;     // bounded: no allocation a file can blow up
	int channels = (ctype == 0) ? 1 : (ctype == 2) ? 3 : (ctype == 3) ? 1 : (ctype == 4) ? 2 : (ctype == 6) ? 4 : 0;
	if (!channels || !(depth == 8 || ((ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4)))) return false;
	size_t bitsPerPixel = (size_t)channels * depth;
	size_t stride = (w * bitsPerPixel + 7) / 8, bpp = std::max<size_t>(1, bitsPerPixel / 8);
	std::vector<uint8_t> raw((stride + 1) * h);
	uLongf rawLen = (uLongf)raw.size();
	if (uncompress(raw.data(), &rawLen, idat.data(), (uLong)idat.size()) != Z_OK || rawLen != raw.size()) return false;
	std::vector<uint8_t> img(stride * h);
	for (uint32_t y = 0; y < h; y++) {
		const uint8_t* src = &raw[(stride + 1) * y];
		uint8_t* dst = &img[stride * y];
		const uint8_t* up = y ? &img[stride * (y - 1)] : nullptr;
		int ft = src[0]; src++;
		for (size_t x = 0; x < stride; x++) {
			int a = x >= bpp ? dst[x - bpp] : 0, b = up ? up[x] : 0, c = (up && x >= bpp) ? up[x - bpp] : 0, v = src[x];
			switch (ft) {
			case 1: v += a; break;
			case 2: v += b; break;
			case 3: v += (a + b) >> 1; break;
			case 4: { int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c); v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
			default: break;
			}
			dst[x] = (uint8_t)v;
		}
	}
	out.resize((size_t)w * h);
	for (uint32_t y = 0; y < h; y++)
		for (uint32_t x = 0; x < w; x++) {
			const uint8_t* row = &img[stride * y];
			uint32_t r, g, b;
			if (depth < 8) {
				size_t bit = (size_t)x * depth;
				int v = (row[bit >> 3] >> (8 - depth - (bit & 7))) & ((1 << depth) - 1);
				if (ctype == 3) { r = pal[v][0]; g = pal[v][1]; b = pal[v][2]; }
				else { r = g = b = (uint32_t)(v * 255 / ((1 << depth) - 1)); }
			} else {
				const uint8_t* p = &row[(size_t)x * channels];
				if (ctype == 3) { r = pal[p[0]][0]; g = pal[p[0]][1]; b = pal[p[0]][2]; }
				else if (channels <= 2) { r = g = b = p[0]; }
				else { r = p[0]; g = p[1]; b = p[2]; }
			}
			out[(size_t)y * w + x] = b | (g << 8) | (r << 16) | (255u << 24);   // MainFrm.cpp:265-268
		}
	width = (int32_t)w; height = (int32_t)h;
	return true;
}

// A file that cannot be decoded, for whatever reason, is a `false`, never an exception: the callers sit
// behind a C ABI.
bool LoadPng(const char* path, std::vector<uint32_t>& out, int32_t& width, int32_t& height)
{
	try { return LoadPngUnguarded(path, out, width, height); }
	catch (...) { out.clear(); return false; }
}

} // namespace pth
