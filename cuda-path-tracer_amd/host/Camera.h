// First-person camera of the headless harness.
//
// Keeps the member interface the reference's callers use (Camera.h / Camera.cpp:
// SetPosition, SetLens, RotateY, Pitch, UpdateViewMatrix, View, Proj) on top of a small
// row-major, row-vector 4x4 type instead of DirectXMath, which is not available
// outside the Windows SDK.  sin/cos come from libm; DirectXMath's XMScalarSinCos
// polynomial is not reproduced, so matrices can differ from the reference's in the
// last bits.  They are inputs to the render call, pinned by value: every renderer that is
// compared receives the same numbers.
#pragma once
#include <cmath>

namespace pth {

struct Float3 { float x, y, z; };

struct Float4x4 {
	float m[4][4];
	float operator()(int r, int c) const { return m[r][c]; }
	float& operator()(int r, int c) { return m[r][c]; }
};

Float4x4 MatrixIdentity();
Float4x4 MatrixMultiply(const Float4x4& a, const Float4x4& b);
// general inverse by cofactors (what XMMatrixInverse computes); returns false if singular
bool MatrixInverse(const Float4x4& a, Float4x4& out);
// XMMatrixPerspectiveFovLH: (0,0) = cot(fovY/2)/aspect, (1,1) = cot(fovY/2)
Float4x4 MatrixPerspectiveFovLH(float fovY, float aspect, float zn, float zf);
// XMMatrixRotationY / XMMatrixRotationAxis for row vectors
Float4x4 MatrixRotationY(float angle);
Float4x4 MatrixRotationAxis(Float3 axis, float angle);
Float3 TransformNormal(Float3 v, const Float4x4& m);

class Camera
{
public:
	Camera();                                               // Camera.cpp:7-15
	void SetPosition(float x, float y, float z);            // Camera.cpp:31-34
	Float3 GetPosition() const { return mPosition; }
	Float3 GetLook() const { return mLook; }
	void SetLens(float fovY, float aspect, float zn, float zf); // Camera.cpp:117-130
	void Pitch(float angle);                                // Camera.cpp:226-234
	void RotateY(float angle);                              // Camera.cpp:236-245
	void UpdateViewMatrix();                                // Camera.cpp:247-289
	const Float4x4& View() const { return mView; }
	const Float4x4& Proj() const { return mProj; }

private:
	Float3 mPosition, mRight, mUp, mLook;
	float mNearZ, mFarZ, mAspect, mFovY;
	Float4x4 mView, mProj;
};

} // namespace pth
