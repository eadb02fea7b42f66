// Wavefront OBJ / MTL reader of the headless harness.
//
// Provides what the reference's scene assembly (MainFrm.cpp:228-486) takes from its
// loader (ObjLoader.h:19-147): positions, normals, texture coordinates, faces with
// per-corner indices, `usemtl` groups as face ranges, and the materials of the
// `mtllib` file (Kd, map_Kd, map_Bump ...).  Plain char paths and std containers; no
// TCHAR, no Windows calls.
#pragma once
#include <string>
#include <vector>

namespace pth {

struct ObjMaterial                // fields of TObjMaterial, ObjLoader.h:19-63, with the same defaults
{
	std::string name = "default";
	float Ka[3] = { 0.2f, 0.2f, 0.2f };
	float Kd[3] = { 0.8f, 0.8f, 0.8f };
	float Ks[3] = { 0.f, 0.f, 0.f };
	float Tf[3] = { 1.f, 1.f, 1.f };
	float Tr = 1.f, Ns = 32.f, Ni = 1.f;
	int illum = 2;
	std::string map_Ka, map_Kd, map_Ks, map_Ns, map_Tr, map_Disp, map_Bump, map_Refl;
};

struct ObjFace                    // TObjMesh::TFace, ObjLoader.h:73-79
{
	int firstVertex = 0;          // into faceVertices
	int firstTexCoord = -1;       // into faceTexCoords, -1 = none
	int firstNormal = -1;         // into faceNormals, -1 = none
	int vCount = 0;
};

struct ObjGroup                   // TObjMesh::TGroup, ObjLoader.h:81-86
{
	unsigned firstFace = 0, numFaces = 0;
	std::string name;
};

struct ObjMesh                    // TObjMesh, ObjLoader.h:69-126
{
	std::vector<float> vertices;      // xyz
	std::vector<float> normals;       // xyz
	std::vector<float> texCoords;     // uv
	std::vector<ObjFace> faces;
	std::vector<int> faceVertices, faceNormals, faceTexCoords;   // 0-based
	std::vector<ObjGroup> matGroups;  // one per `usemtl` statement
	std::vector<ObjMaterial> materials;
	std::string mtlFileName;
	unsigned numTriangles = 0;
	float bbmin[3] = { 0, 0, 0 }, bbmax[3] = { 0, 0, 0 };
};

// Return values as LoadObj, ObjLoader.cpp:210-212: 0 failure, 1 success, 2 loaded but the
// material library was not found.
int LoadObj(const char* fileName, ObjMesh* out);
int LoadMtlLib(const char* fileName, std::vector<ObjMaterial>& materials);

} // namespace pth
