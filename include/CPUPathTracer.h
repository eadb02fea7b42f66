// Drop-in C++ surface of the reference's second render call, the CPU tracer's.
//
// Class name, method name and argument order follow the reference's CPUPathTracer.h:17-18, so that
// its caller (CRayTracerView::CalcRayCPU, RayTracerView.cpp:948-1041, call at :980) compiles against
// this header.  Note the order nDiv, nNumSamples: it is the reverse of CCUDAPathTracer::CalcRays
// (PathTracer.h:48-49).  The output is one DWORD per pixel, B | G << 8 | R << 16 | 255 << 24 with
// every channel truncated from 255 * rgb (CPUPathTracer.cpp:509), row-major h * imageWidth + j.
//
// The implementation (cuda-path-tracer_amd/host/CPUPathTracer.cpp) renders on the GPU through the
// C ABI of include/mipt.h; nothing of the reference's CPU tracer remains behind this call.
//
// Two things the reference takes from its environment are made explicit here:
//  - DirectX::XMMATRIX and CLight come from DirectXMath.h and framework.h:89-100 there.  A caller
//    that has those headers includes them first and defines MIPT_HAVE_REFERENCE_TYPES; otherwise
//    the layout-compatible stand-ins below are used (XMMATRIX: 16 floats, row-major, row-vector
//    convention; CLight: the same five members with the same defaults).
//  - the reference's tracer reads the scene from the global gGlobalData->m_vMeshes
//    (CPUPathTracer.cpp:266).  Here the caller hands it over once with SetScene().
#pragma once

#include <stdint.h>
#include "mipt.h"

#ifndef MIPT_HAVE_REFERENCE_TYPES
#ifndef _WINDEF_
typedef uint32_t DWORD;             // 32 bits on the reference's platform
#endif
namespace DirectX {
struct XMFLOAT3 { float x, y, z; };
struct alignas(16) XMMATRIX { float m[4][4]; };
}
class CLight                        // framework.h:89-100
{
public:
	CLight() {}
	DirectX::XMFLOAT3 colour = { 1, 1, 1 };
	DirectX::XMFLOAT3 direction = { 0, 1, 0 };
	DirectX::XMFLOAT3 pos = { 0, 0, 0 };
	float nScalar = 3000;
	float nIntensity = 3;
};
#endif

class CPTCallback;

class CCPUPathTracer
{
public:
	// Stands in for gGlobalData->m_vMeshes: the meshes (one per material group, DWORD texels =
	// MIPT_TEX_BGRA8, VBVertex = mipt_vertex) the next CalcRays calls render.  The descriptor and
	// everything it points to must stay valid until the last of those calls has returned; the
	// device copy and its BVH are made by the first CalcRays after SetScene and kept for later
	// calls.  SetScene(nullptr) releases them.
	static void SetScene(const mipt_scene_desc* pScene);

	// CPUPathTracer.h:17-18.  Synchronous; the callback runs on the calling thread, first
	// (0, total), then once per pass of samples up to (total, total) - the reference calls it once
	// per pixel (CPUPathTracer.cpp:508).  sun.pos and sun.nIntensity are used, as in the reference
	// (CPUPathTracer.cpp:373-375, 405).  On failure a message goes to stderr and pOutputImage is
	// left untouched.
	void CalcRays(CPTCallback* pCallback, DWORD* pOutputImage, long nClientWidth, long nClientHeight, long nDiv, long nNumSamples, float P0, float P1, DirectX::XMMATRIX ToLocal,
		CLight sun, bool bGlobalIllumination, bool bUseTextures, bool bUseNormals);
};
