// Wavefront OBJ / MTL reader (see ObjLoader.h).  One pass over the file; the statements
// understood are those the reference's loader acts on (ObjLoader.cpp:241-430 for OBJ,
// :144-176 for MTL): v, vn, vt, f, usemtl, mtllib / newmtl, Ka, Kd, Ks, Tf, Tr, d, Ns,
// Ni, illum, map_*.
#include "ObjLoader.h"

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <strings.h>

namespace pth {

static std::string FirstToken(const char* s)
{
	while (*s == ' ' || *s == '\t') s++;
	const char* e = s;
	while (*e && !isspace((unsigned char)*e)) e++;
	return std::string(s, e);
}

static bool StartsWithNoCase(const char* line, const char* key)
{
	return strncasecmp(line, key, strlen(key)) == 0;
}

// Ka/Kd/Ks/Tf: one value means grey (ObjLoader.cpp:18-27); "spectral" and "xyz" forms are skipped
static void ReadColour(const char* args, float* k)
{
	std::string first = FirstToken(args);
	if (strcasecmp(first.c_str(), "spectral") == 0 || strcasecmp(first.c_str(), "xyz") == 0) return;
	float a, b, c;
	int n = sscanf(args, "%f %f %f", &a, &b, &c);
	if (n == 3) { k[0] = a; k[1] = b; k[2] = c; }
	else if (n >= 1) { k[0] = k[1] = k[2] = a; }
}

int LoadMtlLib(const char* fileName, std::vector<ObjMaterial>& materials)
{
	FILE* f = fopen(fileName, "r");
	if (!f) return 0;
	char line[4096];
	ObjMaterial* cur = nullptr;
	while (fgets(line, sizeof(line), f)) {
		const char* s = line;
		while (*s == ' ' || *s == '\t') s++;
		if (strncmp(s, "newmtl ", 7) == 0) {
			materials.emplace_back();
			cur = &materials.back();
			cur->name = FirstToken(s + 7);
			continue;
		}
		if (!cur) continue;
		if (StartsWithNoCase(s, "ka ")) ReadColour(s + 3, cur->Ka);
		else if (StartsWithNoCase(s, "ks ")) ReadColour(s + 3, cur->Ks);
		else if (StartsWithNoCase(s, "kd ")) ReadColour(s + 3, cur->Kd);
		else if (StartsWithNoCase(s, "tf ")) ReadColour(s + 3, cur->Tf);
		else if (StartsWithNoCase(s, "tr ")) cur->Tr = (float)atof(s + 3);
		else if (StartsWithNoCase(s, "d ")) cur->Tr = (float)atof(s + 2);
		else if (StartsWithNoCase(s, "ns ")) cur->Ns = (float)atof(s + 3);
		else if (StartsWithNoCase(s, "ni ")) cur->Ni = (float)atof(s + 3);
		else if (StartsWithNoCase(s, "illum ")) cur->illum = atoi(s + 6);
		else if (StartsWithNoCase(s, "map_ka ")) cur->map_Ka = FirstToken(s + 7);
		else if (StartsWithNoCase(s, "map_kd ")) cur->map_Kd = FirstToken(s + 7);
		else if (StartsWithNoCase(s, "map_ks ")) cur->map_Ks = FirstToken(s + 7);
		else if (StartsWithNoCase(s, "map_ns ")) cur->map_Ns = FirstToken(s + 7);
		else if (StartsWithNoCase(s, "map_tr ")) cur->map_Tr = FirstToken(s + 7);
		else if (StartsWithNoCase(s, "map_disp ")) cur->map_Disp = FirstToken(s + 9);
		else if (StartsWithNoCase(s, "map_bump ")) cur->map_Bump = FirstToken(s + 9);
		else if (StartsWithNoCase(s, "map_refl ")) cur->map_Refl = FirstToken(s + 9);
	}
	fclose(f);
	return 1;
}

// one face corner "v", "v/t", "v//n" or "v/t/n"; returns false if there is no position index
static bool ParseCorner(const char* tok, int& v, int& t, int& n, bool& hasT, bool& hasN)
{
	v = t = n = 0; hasT = hasN = false;
	char* end = nullptr;
	long iv = strtol(tok, &end, 10);
	if (end == tok) return false;
	v = (int)iv;
	if (*end != '/') return true;
	const char* p = end + 1;
	long it = strtol(p, &end, 10);
	if (end != p) { t = (int)it; hasT = true; }
	if (*end != '/') return true;
	p = end + 1;
	long in = strtol(p, &end, 10);
	if (end != p) { n = (int)in; hasN = true; }
	return true;
}

int LoadObj(const char* fileName, ObjMesh* out)
{
	FILE* f = fopen(fileName, "r");
	if (!f) return 0;
	ObjMesh& obj = *out;
	obj = ObjMesh();
	char line[4096];
	bool firstFace = true, fileHasT = false, fileHasN = false;
	while (fgets(line, sizeof(line), f)) {
		if (strncmp(line, "v ", 2) == 0) {
			float x = 0, y = 0, z = 0;
			sscanf(line + 1, "%f %f %f", &x, &y, &z);
			if (obj.vertices.empty()) { obj.bbmin[0] = obj.bbmax[0] = x; obj.bbmin[1] = obj.bbmax[1] = y; obj.bbmin[2] = obj.bbmax[2] = z; }
			const float p[3] = { x, y, z };
			for (int a = 0; a < 3; a++) {
				if (p[a] < obj.bbmin[a]) obj.bbmin[a] = p[a]; else if (p[a] > obj.bbmax[a]) obj.bbmax[a] = p[a];
			}
			obj.vertices.insert(obj.vertices.end(), { x, y, z });
		} else if (strncmp(line, "vn ", 3) == 0) {
			float x = 0, y = 0, z = 0;
			sscanf(line + 2, "%f %f %f", &x, &y, &z);
			obj.normals.insert(obj.normals.end(), { x, y, z });
		} else if (strncmp(line, "vt ", 3) == 0) {
			float u = 0, v = 0;
			sscanf(line + 2, "%f %f", &u, &v);
			obj.texCoords.insert(obj.texCoords.end(), { u, v });
		} else if (strncmp(line, "f ", 2) == 0) {
			ObjFace face;
			face.firstVertex = (int)obj.faceVertices.size();
			const char* s = line + 1;
			int count = 0;
			while (*s) {
				while (*s == ' ' || *s == '\t') s++;
				if (!*s || *s == '\n' || *s == '\r') break;
				int v, t, n; bool hasT, hasN;
				if (!ParseCorner(s, v, t, n, hasT, hasN)) break;
				if (firstFace) {      // which components exist is decided by the first face of the file
					fileHasT = hasT; fileHasN = hasN; firstFace = false;  // (ObjLoader.cpp:255)
				}
				if (count == 0) {
					face.firstTexCoord = fileHasT ? (int)obj.faceTexCoords.size() : -1;
					face.firstNormal = fileHasN ? (int)obj.faceNormals.size() : -1;
				}
				int nv = (int)(obj.vertices.size() / 3), nt = (int)(obj.texCoords.size() / 2), nn = (int)(obj.normals.size() / 3);
				// negative indices count back from the latest element
				obj.faceVertices.push_back(v < 0 ? nv + v : v - 1);
				if (fileHasT) obj.faceTexCoords.push_back(hasT ? (t < 0 ? nt + t : t - 1) : 0);
				if (fileHasN) obj.faceNormals.push_back(hasN ? (n < 0 ? nn + n : n - 1) : 0);
				count++;
				while (*s && !isspace((unsigned char)*s)) s++;
			}
			face.vCount = count;
			if (count >= 3) {
				obj.numTriangles += (unsigned)(count - 2);
				obj.faces.push_back(face);
			} else {
				obj.faceVertices.resize((size_t)face.firstVertex);
				if (fileHasT && face.firstTexCoord >= 0) obj.faceTexCoords.resize((size_t)face.firstTexCoord);
				if (fileHasN && face.firstNormal >= 0) obj.faceNormals.resize((size_t)face.firstNormal);
			}
		} else if (StartsWithNoCase(line, "usemtl ")) {
			unsigned fc = (unsigned)obj.faces.size();
			if (!obj.matGroups.empty()) obj.matGroups.back().numFaces = fc - obj.matGroups.back().firstFace;
			ObjGroup g;
			g.firstFace = fc;
			g.name = FirstToken(line + 7);
			obj.matGroups.push_back(g);
		} else if (StartsWithNoCase(line, "mtllib ")) {
			obj.mtlFileName = FirstToken(line + 7);
		}
	}
	fclose(f);
	if (!obj.matGroups.empty())
		obj.matGroups.back().numFaces = (unsigned)obj.faces.size() - obj.matGroups.back().firstFace;
	if (obj.vertices.empty() || obj.faces.empty()) return 0;
	if (obj.mtlFileName.empty()) return 1;
	// the library name is tried as given, then relative to the OBJ file (ObjLoader.cpp:455-466)
	if (LoadMtlLib(obj.mtlFileName.c_str(), obj.materials)) return 1;
	std::string path(fileName);
	size_t slash = path.find_last_of("/\\");
	std::string dir = (slash == std::string::npos) ? std::string() : path.substr(0, slash + 1);
	if (LoadMtlLib((dir + obj.mtlFileName).c_str(), obj.materials)) return 1;
	return 2;
}

} // namespace pth
