// Headless harness: loads or generates a scene, positions camera and sun the way the
// reference's GUI does (SideView.cpp:144-164 -> RayTracerView.cpp:783-946), renders through
// CCUDAPathTracer::CalcRays (--entry gpu, the default: what CalcRayCUDA calls) or through
// CCPUPathTracer::CalcRays (--entry cpu: what CalcRayCPU calls, RayTracerView.cpp:948-1041) and
// writes a PPM.
//
//   pt_headless --scene cornell|sponza|crytek [--obj file.mesh] [--tris N] [--tex N]
//               [--samples N] [--gi 0|1] [--textures 0|1] [--normals 0|1] [--div 1|2|4]
//               [--width W --height H] [--entry gpu|cpu] [--out image.ppm|image.png]
#include "../../include/PathTracer.h"
#include "../../include/CPUPathTracer.h"
#include "../../include/PTProgress.h"
#include "../../include/mipt_host.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

class ConsoleProgress : public CPTCallback
{
public:
	void UpdateProgress(long nProgress, long nTotal) override
	{
		fprintf(stderr, "\rprogress %ld / %ld", nProgress, nTotal);
		if (nProgress >= nTotal) fprintf(stderr, "\n");
	}
};

int main(int argc, char** argv)
{
	std::string sceneName = "cornell", obj, out = "_ImageHIP.ppm", entry = "gpu";
	long width = 1920, height = 1080, samples = 16, div = 1, tris = 0, tex = 1024;
	bool gi = true, textures = true, normals = false;
	for (int i = 1; i + 1 < argc; i += 2) {
		std::string k = argv[i]; const char* v = argv[i + 1];
		if (k == "--scene") sceneName = v; else if (k == "--obj") obj = v; else if (k == "--out") out = v;
		else if (k == "--width") width = atol(v); else if (k == "--height") height = atol(v);
		else if (k == "--samples") samples = atol(v); else if (k == "--div") div = atol(v);
		else if (k == "--tris") tris = atol(v); else if (k == "--tex") tex = atol(v);
		else if (k == "--gi") gi = atoi(v) != 0; else if (k == "--textures") textures = atoi(v) != 0;
		else if (k == "--normals") normals = atoi(v) != 0;
		else if (k == "--entry") entry = v;
		else { fprintf(stderr, "unknown option %s\n", k.c_str()); return 2; }
	}
	int kind = sceneName == "cornell" ? MIPT_SCENE_CORNELL : sceneName == "sponza" ? MIPT_SCENE_SPONZA : MIPT_SCENE_CRYTEK;
	mipt_host_scene* hs = nullptr;
	if (!obj.empty()) hs = mipt_host_scene_load_obj(obj.c_str(), kind == MIPT_SCENE_CORNELL);
	else if (kind != MIPT_SCENE_CORNELL) hs = mipt_host_scene_atrium(kind, tris ? tris : (kind == MIPT_SCENE_SPONZA ? 66000 : 262000), (int)tex, 1);
	if (!hs) { fprintf(stderr, "no scene: give --obj <file> (the Cornell box needs its .mesh file)\n"); return 1; }
	const mipt_scene_desc* d = mipt_host_scene_desc(hs);
	fprintf(stderr, "scene %s: %lld triangles, %d meshes, %d materials\n", mipt_host_scene_label(hs), (long long)d->num_triangles, d->num_meshes, d->num_materials);

	// marshal as CRayTracerView::CalcRayCUDA does
	if (samples < 1) samples = 1;
	if (samples > 10000) samples = 10000;          // RayTracerView.cpp:785-786
	mipt_camera cam; float sun[3];
	if (mipt_host_camera_preset(kind, (int32_t)width, (int32_t)height, (int32_t)div, &cam, sun) != 0) return 1;
	long W = width / div, H = height / div;
	std::vector<float3> image((size_t)W * H, float3{ 0, 0, 0 });
	ConsoleProgress progress;
	if (entry == "cpu") {
		// marshal as CRayTracerView::CalcRayCPU does (RayTracerView.cpp:948-981): the scene is the
		// global mesh list with its DWORD texels, the light a CLight
		std::vector<DWORD> dwImage((size_t)W * H, 0);
		DirectX::XMMATRIX toLocalM;
		memcpy(&toLocalM, cam.to_local, sizeof(toLocalM));
		CLight sunLight;
		sunLight.pos = { sun[0], sun[1], sun[2] };
		CCPUPathTracer::SetScene(d);
		CCPUPathTracer PT;
		auto t0 = std::chrono::steady_clock::now();
		PT.CalcRays(&progress, dwImage.data(), width, height, div, samples, cam.p0, cam.p1, toLocalM, sunLight, gi, textures, normals);
		double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		fprintf(stderr, "rendered %ldx%ld in %.3f s (CCPUPathTracer entry)\n", W, H, sec);
		CCPUPathTracer::SetScene(nullptr);
		// the bytes of the DWORD image, as CalcRayCPU copies them into its CImage (RayTracerView.cpp:1017-1026)
		for (size_t i = 0; i < dwImage.size(); i++)
			image[i] = float3{ (float)(dwImage[i] & 255u), (float)((dwImage[i] >> 8) & 255u), (float)((dwImage[i] >> 16) & 255u) };
		if (out.size() > 4 && out.compare(out.size() - 4, 4, ".png") == 0)
			mipt_host_write_png(out.c_str(), (const float*)image.data(), (int32_t)W, (int32_t)H);
		else
			mipt_host_write_ppm(out.c_str(), (const float*)image.data(), (int32_t)W, (int32_t)H);
		mipt_host_scene_free(hs);
		return 0;
	}
	if (entry != "gpu") { fprintf(stderr, "unknown entry %s (gpu or cpu)\n", entry.c_str()); return 2; }
	std::vector<CUDAMaterial> mats((size_t)d->num_materials);
	std::vector<std::vector<float3>> texels((size_t)d->num_materials * 2);
	auto unpack = [](const void* data, int w, int h, std::vector<float3>& dst) {
		const uint32_t* px = (const uint32_t*)data;
		dst.resize((size_t)w * h);
		for (size_t i = 0; i < dst.size(); i++) dst[i] = { (float)(px[i] & 255u), (float)((px[i] >> 8) & 255u), (float)((px[i] >> 16) & 255u) };
	};
	for (int i = 0; i < d->num_materials; i++) {    // RayTracerView.cpp:813-863
		const mipt_material& m = d->materials[i];
		CUDAMaterial& c = mats[(size_t)i];
		if (m.tex_data) { unpack(m.tex_data, m.tex_width, m.tex_height, texels[2 * (size_t)i]); c.pTexData = texels[2 * (size_t)i].data(); c.nWidth = m.tex_width; c.nHeight = m.tex_height; }
		if (m.nml_data) { unpack(m.nml_data, m.nml_width, m.nml_height, texels[2 * (size_t)i + 1]); c.pNmlData = texels[2 * (size_t)i + 1].data(); c.nNmlWidth = m.nml_width; c.nNmlHeight = m.nml_height; }
		c.diffuse = { m.diffuse[0], m.diffuse[1], m.diffuse[2] };
	}
	std::vector<CUDAMesh> meshes((size_t)d->num_meshes);
	for (int i = 0; i < d->num_meshes; i++) {       // RayTracerView.cpp:867-882
		const mipt_mesh& m = d->meshes[i];
		CUDAMesh& c = meshes[(size_t)i];
		c.pVertices = (CCUDAVertex*)&d->vertices[3 * m.first_triangle];
		c.nNumTriangles = (long)m.num_triangles;
		c.bLight = m.is_light != 0;
		c.pMaterial = m.material >= 0 ? &mats[(size_t)m.material] : nullptr;
	}
	float toLocal[4][4];
	memcpy(toLocal, cam.to_local, sizeof(toLocal));
	auto t0 = std::chrono::steady_clock::now();
	CCUDAPathTracer::CalcRays(&progress, image.data(), width, height, samples, div, cam.p0, cam.p1, toLocal,
		float3{ sun[0], sun[1], sun[2] }, float3{ 0, 1, 0 }, 3.0f, gi, textures, normals, meshes.data(), (long)meshes.size(), mats.data(), (long)mats.size());
	double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	fprintf(stderr, "rendered %ldx%ld in %.3f s\n", W, H, sec);
	if (out.size() > 4 && out.compare(out.size() - 4, 4, ".png") == 0)
		mipt_host_write_png(out.c_str(), (const float*)image.data(), (int32_t)W, (int32_t)H);
	else
		mipt_host_write_ppm(out.c_str(), (const float*)image.data(), (int32_t)W, (int32_t)H);
	mipt_host_scene_free(hs);
	return 0;
}
