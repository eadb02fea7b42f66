// CCPUPathTracer::CalcRays on top of the C ABI.
//
// Same signature and calling contract as the reference's CPUPathTracer.cpp:474-531 (synchronous,
// DWORD image owned by the caller, callback on the calling thread), but the image is rendered on
// the GPU by libmipt.so and packed to DWORD by the harness's output stage.  The scene, which the
// reference reads from a global (CPUPathTracer.cpp:266), is the one given to SetScene().
#include "../../include/CPUPathTracer.h"
#include "../../include/PTProgress.h"
#include "../../include/mipt.h"
#include "../../include/mipt_host.h"

#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

namespace {
std::mutex gLock;
const mipt_scene_desc* gDesc = nullptr;
mipt_scene* gScene = nullptr;           // device copy + BVH of gDesc, made on first use

struct ProgressThunk { CPTCallback* cb; };
void ProgressTrampoline(int64_t done, int64_t total, void* user)
{
	ProgressThunk* t = (ProgressThunk*)user;
	if (t && t->cb) t->cb->UpdateProgress((long)done, (long)total);
}
}

void CCPUPathTracer::SetScene(const mipt_scene_desc* pScene)
{
	std::lock_guard<std::mutex> g(gLock);
	if (gScene) { mipt_scene_destroy(gScene); gScene = nullptr; }
	gDesc = pScene;
}

void CCPUPathTracer::CalcRays(CPTCallback* pCallback, DWORD* pOutputImage, long nClientWidth, long nClientHeight, long nDiv, long nNumSamples, float P0, float P1, DirectX::XMMATRIX ToLocal,
	CLight sun, bool bGlobalIllumination, bool bUseTextures, bool bUseNormals)
{
	std::lock_guard<std::mutex> g(gLock);
	if (!gDesc) { fprintf(stderr, "CCPUPathTracer::CalcRays: no scene, call SetScene first\n"); return; }
	if (!pOutputImage || nDiv <= 0) { fprintf(stderr, "CCPUPathTracer::CalcRays: bad arguments\n"); return; }
	if (!gScene) {
		if (mipt_init(0) != MIPT_OK) { fprintf(stderr, "mipt_init failed: %s\n", mipt_last_error()); return; }
		if (mipt_scene_create(gDesc, nullptr, &gScene) != MIPT_OK) {
			fprintf(stderr, "mipt_scene_create failed: %s\n", mipt_last_error());
			gScene = nullptr;
			return;
		}
	}
	mipt_camera cam = {};
	cam.client_width = (int32_t)nClientWidth; cam.client_height = (int32_t)nClientHeight; cam.div = (int32_t)nDiv;
	cam.p0 = P0; cam.p1 = P1;
	static_assert(sizeof(DirectX::XMMATRIX) == 16 * sizeof(float), "XMMATRIX is 16 floats, row-major");
	memcpy(cam.to_local, &ToLocal, sizeof(cam.to_local));

	mipt_params prm;
	mipt_host_default_params(&prm);          // epsilon 1e-20, normal map not normalised: the CPU flavour
	prm.num_samples = (int32_t)nNumSamples;
	prm.sun_pos[0] = sun.pos.x; prm.sun_pos[1] = sun.pos.y; prm.sun_pos[2] = sun.pos.z;
	prm.sun_intensity = sun.nIntensity;
	prm.global_illumination = bGlobalIllumination; prm.use_textures = bUseTextures; prm.use_normals = bUseNormals;

	const long W = nClientWidth / nDiv, H = nClientHeight / nDiv;       // CPUPathTracer.cpp:478-479
	if (W <= 0 || H <= 0) { fprintf(stderr, "CCPUPathTracer::CalcRays: empty image\n"); return; }
	std::vector<float> bgr((size_t)W * (size_t)H * 3);
	ProgressThunk thunk{ pCallback };
	if (mipt_render(gScene, &cam, &prm, bgr.data(), nullptr, nullptr, pCallback ? ProgressTrampoline : nullptr, &thunk) != MIPT_OK) {
		fprintf(stderr, "mipt_render failed: %s\n", mipt_last_error());
		return;
	}
	mipt_host_pack_bgra(bgr.data(), (int64_t)W * H, (uint32_t*)pOutputImage);      // CPUPathTracer.cpp:509
}
