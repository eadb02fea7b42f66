// Scene container and scene assembly of the headless harness.
//
// HostScene is the headless counterpart of the reference's CGlobalData (framework.h:
// 167-179): the list of meshes (one per material group) and the list of materials, held
// as the plain arrays the render call takes (include/mipt.h).
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include "../../include/mipt.h"
#include "../../include/mipt_host.h"

namespace pth {

struct HostMaterial               // CTexture, framework.h:73-87
{
	std::string name;
	std::vector<uint32_t> tex;    // BGRA8 DWORD texels, byte0 = blue
	int32_t width = 0, height = 0;
	std::vector<uint32_t> nml;
	int32_t nmlWidth = 0, nmlHeight = 0;
	float diffuse[3] = { 0, 0, 0 };   // BGR (MainFrm.cpp:247)
};

struct HostMesh                   // the render-relevant part of CMesh, framework.h:102-158
{
	std::string material;
	int32_t materialIndex = -1;
	bool light = false;
	std::vector<mipt_vertex> vertices;   // 3 per triangle
};

class HostScene
{
public:
	std::string label;
	std::vector<HostMesh> meshes;
	std::vector<HostMaterial> materials;

	// Orders the meshes by triangle count descending (MainFrm.cpp:489-492, stable here),
	// binds materials by name (MainFrm.cpp:445-457) and builds the flat descriptor.
	void Finalize(bool sortMeshes);
	const mipt_scene_desc* Desc() const { return &mDesc; }
	int64_t NumTriangles() const { return mDesc.num_triangles; }

private:
	std::vector<mipt_vertex> mVertices;
	std::vector<mipt_mesh> mMeshDesc;
	std::vector<mipt_material> mMatDesc;
	mipt_scene_desc mDesc = {};
};

// MainFrm.cpp:228-486 without Direct3D / WIC.
HostScene* AssembleObjScene(const char* path, bool flipWinding);
// Procedural stand-in for the Sponza meshes that the reference tree does not contain.
HostScene* BuildAtrium(int sceneKind, int64_t targetTriangles, int textureSize, uint32_t seed);
HostScene* BuildInstanced(const HostScene& base, int copies, float spacing);

// PNG (8-bit grey / RGB / RGBA / palette, 1-bit grey; non-interlaced) -> BGRA8 DWORDs
bool LoadPng(const char* path, std::vector<uint32_t>& out, int32_t& width, int32_t& height);
// baseline JPEG (Jpeg.cpp); same texel layout
bool LoadJpeg(const char* path, std::vector<uint32_t>& out, int32_t& width, int32_t& height);
// 8-bit RGB PNG from BGRA8 DWORDs (alpha dropped), zlib-compressed, filter type 0
bool WritePng(const char* path, const std::vector<uint32_t>& bgra, int32_t width, int32_t height);
// PNG or JPEG by the file's first bytes
bool LoadImageFile(const char* path, std::vector<uint32_t>& out, int32_t& width, int32_t& height);

} // namespace pth
