// CCUDAPathTracer::CalcRays on top of the C ABI.
//
// Same signature and calling contract as the reference's PathTracer.cu:556-771
// (synchronous, caller owns every host array, callback on the calling thread), but the
// body only flattens the caller's mesh array into the plain arrays of include/mipt.h and
// calls mipt_scene_create / mipt_render / mipt_scene_destroy.
#include "../../include/PathTracer.h"
#include "../../include/PTProgress.h"
#include "../../include/mipt.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace {
struct ProgressThunk { CPTCallback* cb; };
void ProgressTrampoline(int64_t done, int64_t total, void* user)
{
	ProgressThunk* t = (ProgressThunk*)user;
	if (t && t->cb) t->cb->UpdateProgress((long)done, (long)total);
}
}

void CCUDAPathTracer::CalcRays(CPTCallback* pCallback, float3* pOutputImage, long nClientWidth, long nClientHeight, long nNumSamples, long nDiv, float P0, float P1, float ToLocal[4][4],
	float3 nSunPos, float3 /*nSunDir: unused by the reference kernel too*/, float nSunIntensity, bool bGlobalIllumination, bool bUseTextures, bool bUseNormals, CUDAMesh* pVB, long nNumMeshs, CUDAMaterial* pTextures, long nNumTextures)
{
	static_assert(sizeof(CCUDAVertex) == sizeof(mipt_vertex), "vertex layouts must match");
	using clk = std::chrono::steady_clock;
	auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
	const clk::time_point t0 = clk::now();
	std::vector<mipt_material> mats((size_t)nNumTextures);
	for (long i = 0; i < nNumTextures; i++) {
		mipt_material& m = mats[(size_t)i];
		m = mipt_material{};
		const CUDAMaterial& s = pTextures[i];
		if (s.pTexData && s.nWidth > 0 && s.nHeight > 0) {
			m.tex_data = s.pTexData; m.tex_width = (int32_t)s.nWidth; m.tex_height = (int32_t)s.nHeight; m.tex_format = MIPT_TEX_F32X3;
		}
		if (s.pNmlData && s.nNmlWidth > 0 && s.nNmlHeight > 0) {
			m.nml_data = s.pNmlData; m.nml_width = (int32_t)s.nNmlWidth; m.nml_height = (int32_t)s.nNmlHeight; m.nml_format = MIPT_TEX_F32X3;
		}
		m.diffuse[0] = s.diffuse.x; m.diffuse[1] = s.diffuse.y; m.diffuse[2] = s.diffuse.z;
	}
	std::vector<mipt_mesh> meshes((size_t)nNumMeshs);
	std::vector<mipt_vertex> verts;
	size_t total = 0;
	for (long i = 0; i < nNumMeshs; i++) total += (size_t)pVB[i].nNumTriangles * 3;
	verts.reserve(total);
	for (long i = 0; i < nNumMeshs; i++) {
		const CUDAMesh& s = pVB[i];
		mipt_mesh& m = meshes[(size_t)i];
		m.first_triangle = (int64_t)(verts.size() / 3);
		m.num_triangles = s.pVertices ? s.nNumTriangles : 0;
		m.is_light = s.bLight ? 1 : 0;
		m.material = -1;
		if (s.pMaterial)          // the material is identified by its address in the caller's array (PathTracer.cu:668-676)
			for (long j = 0; j < nNumTextures; j++)
				if (s.pMaterial == &pTextures[j]) { m.material = (int32_t)j; break; }
		const mipt_vertex* v = (const mipt_vertex*)s.pVertices;
		verts.insert(verts.end(), v, v + (size_t)m.num_triangles * 3);
	}
	mipt_scene_desc desc = {};
	desc.vertices = verts.data(); desc.num_triangles = (int64_t)(verts.size() / 3);
	desc.meshes = meshes.data(); desc.num_meshes = (int32_t)nNumMeshs;
	desc.materials = mats.data(); desc.num_materials = (int32_t)nNumTextures;

	mipt_camera cam = {};
	cam.client_width = (int32_t)nClientWidth; cam.client_height = (int32_t)nClientHeight; cam.div = (int32_t)nDiv;
	cam.p0 = P0; cam.p1 = P1;
	for (int w = 0; w < 4; w++) for (int q = 0; q < 4; q++) cam.to_local[w * 4 + q] = ToLocal[w][q];

	mipt_params prm = {};
	prm.num_samples = (int32_t)nNumSamples;
	prm.sun_pos[0] = nSunPos.x; prm.sun_pos[1] = nSunPos.y; prm.sun_pos[2] = nSunPos.z;
	prm.sun_intensity = nSunIntensity;
	prm.global_illumination = bGlobalIllumination; prm.use_textures = bUseTextures; prm.use_normals = bUseNormals;
	prm.epsilon = 1e-20f;
	prm.seed = 0x5EED;
	prm.shard_count = 1;

	mipt_scene* scene = nullptr;
	const clk::time_point t1 = clk::now();
	if (mipt_init(0) != MIPT_OK) {     // the reference renders on device 0 (PathTracer.cu:577)
		fprintf(stderr, "mipt_init failed: %s\n", mipt_last_error());
		return;
	}
	const clk::time_point t2 = clk::now();
	if (mipt_scene_create(&desc, nullptr, &scene) != MIPT_OK) {
		fprintf(stderr, "mipt_scene_create failed: %s\n", mipt_last_error());
		return;
	}
	const clk::time_point t3 = clk::now();
	ProgressThunk thunk{ pCallback };
	if (mipt_render(scene, &cam, &prm, (float*)pOutputImage, nullptr, nullptr, pCallback ? ProgressTrampoline : nullptr, &thunk) != MIPT_OK)
		fprintf(stderr, "mipt_render failed: %s\n", mipt_last_error());
	const clk::time_point t4 = clk::now();
	if (getenv("MIPT_VERBOSE")) {
		mipt_scene_info inf = {};
		mipt_scene_get_info(scene, &inf);
		fprintf(stderr, "CalcRays: flatten %.1f ms, device init %.1f ms, scene %.1f ms (upload %.1f, BVH build %.1f), render %.1f ms\n",
		        ms(t0, t1), ms(t1, t2), ms(t2, t3), inf.upload_ms, inf.build_ms, ms(t3, t4));
	}
	mipt_scene_destroy(scene);
}
