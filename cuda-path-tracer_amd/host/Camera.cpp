// Camera and the handful of 4x4 helpers the harness needs (see Camera.h).
#include "Camera.h"

namespace pth {

static inline Float3 Sub(Float3 a, Float3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
static inline float Dot(Float3 a, Float3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline Float3 Cross(Float3 a, Float3 b)
{
	return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x };
}
static inline Float3 Normalize(Float3 v)
{
	float len = std::sqrt(Dot(v, v));
	if (len == 0.0f) return { 0, 0, 0 };
	return { v.x / len, v.y / len, v.z / len };
}

Float4x4 MatrixIdentity()
{
	Float4x4 r = {};
	for (int i = 0; i < 4; i++) r.m[i][i] = 1.0f;
	return r;
}

Float4x4 MatrixMultiply(const Float4x4& a, const Float4x4& b)
{
	Float4x4 r;
	for (int i = 0; i < 4; i++)
		for (int j = 0; j < 4; j++)
			r.m[i][j] = ((a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j]) + a.m[i][2] * b.m[2][j]) + a.m[i][3] * b.m[3][j];
	return r;
}

bool MatrixInverse(const Float4x4& a, Float4x4& out)
{
	// adjugate over determinant, on the flattened matrix
	const float* s = &a.m[0][0];
	float inv[16];
	inv[0] = s[5] * s[10] * s[15] - s[5] * s[11] * s[14] - s[9] * s[6] * s[15] + s[9] * s[7] * s[14] + s[13] * s[6] * s[11] - s[13] * s[7] * s[10];
	inv[4] = -s[4] * s[10] * s[15] + s[4] * s[11] * s[14] + s[8] * s[6] * s[15] - s[8] * s[7] * s[14] - s[12] * s[6] * s[11] + s[12] * s[7] * s[10];
	inv[8] = s[4] * s[9] * s[15] - s[4] * s[11] * s[13] - s[8] * s[5] * s[15] + s[8] * s[7] * s[13] + s[12] * s[5] * s[11] - s[12] * s[7] * s[9];
	inv[12] = -s[4] * s[9] * s[14] + s[4] * s[10] * s[13] + s[8] * s[5] * s[14] - s[8] * s[6] * s[13] - s[12] * s[5] * s[10] + s[12] * s[6] * s[9];
	inv[1] = -s[1] * s[10] * s[15] + s[1] * s[11] * s[14] + s[9] * s[2] * s[15] - s[9] * s[3] * s[14] - s[13] * s[2] * s[11] + s[13] * s[3] * s[10];
	inv[5] = s[0] * s[10] * s[15] - s[0] * s[11] * s[14] - s[8] * s[2] * s[15] + s[8] * s[3] * s[14] + s[12] * s[2] * s[11] - s[12] * s[3] * s[10];
	inv[9] = -s[0] * s[9] * s[15] + s[0] * s[11] * s[13] + s[8] * s[1] * s[15] - s[8] * s[3] * s[13] - s[12] * s[1] * s[11] + s[12] * s[3] * s[9];
	inv[13] = s[0] * s[9] * s[14] - s[0] * s[10] * s[13] - s[8] * s[1] * s[14] + s[8] * s[2] * s[13] + s[12] * s[1] * s[10] - s[12] * s[2] * s[9];
	inv[2] = s[1] * s[6] * s[15] - s[1] * s[7] * s[14] - s[5] * s[2] * s[15] + s[5] * s[3] * s[14] + s[13] * s[2] * s[7] - s[13] * s[3] * s[6];
	inv[6] = -s[0] * s[6] * s[15] + s[0] * s[7] * s[14] + s[4] * s[2] * s[15] - s[4] * s[3] * s[14] - s[12] * s[2] * s[7] + s[12] * s[3] * s[6];
	inv[10] = s[0] * s[5] * s[15] - s[0] * s[7] * s[13] - s[4] * s[1] * s[15] + s[4] * s[3] * s[13] + s[12] * s[1] * s[7] - s[12] * s[3] * s[5];
	inv[14] = -s[0] * s[5] * s[14] + s[0] * s[6] * s[13] + s[4] * s[1] * s[14] - s[4] * s[2] * s[13] - s[12] * s[1] * s[6] + s[12] * s[2] * s[5];
	inv[3] = -s[1] * s[6] * s[11] + s[1] * s[7] * s[10] + s[5] * s[2] * s[11] - s[5] * s[3] * s[10] - s[9] * s[2] * s[7] + s[9] * s[3] * s[6];
	inv[7] = s[0] * s[6] * s[11] - s[0] * s[7] * s[10] - s[4] * s[2] * s[11] + s[4] * s[3] * s[10] + s[8] * s[2] * s[7] - s[8] * s[3] * s[6];
	inv[11] = -s[0] * s[5] * s[11] + s[0] * s[7] * s[9] + s[4] * s[1] * s[11] - s[4] * s[3] * s[9] - s[8] * s[1] * s[7] + s[8] * s[3] * s[5];
	inv[15] = s[0] * s[5] * s[10] - s[0] * s[6] * s[9] - s[4] * s[1] * s[10] + s[4] * s[2] * s[9] + s[8] * s[1] * s[6] - s[8] * s[2] * s[5];
	float det = s[0] * inv[0] + s[1] * inv[4] + s[2] * inv[8] + s[3] * inv[12];
	if (det == 0.0f) return false;
	float r = 1.0f / det;
	float* o = &out.m[0][0];
	for (int i = 0; i < 16; i++) o[i] = inv[i] * r;
	return true;
}

Float4x4 MatrixPerspectiveFovLH(float fovY, float aspect, float zn, float zf)
{
	float s = std::sin(0.5f * fovY), c = std::cos(0.5f * fovY);
	float height = c / s;
	float width = height / aspect;
	float range = zf / (zf - zn);
	Float4x4 r = {};
	r.m[0][0] = width;
	r.m[1][1] = height;
	r.m[2][2] = range;
	r.m[2][3] = 1.0f;
	r.m[3][2] = -range * zn;
	return r;
}

Float4x4 MatrixRotationY(float angle)
{
	float s = std::sin(angle), c = std::cos(angle);
	Float4x4 r = MatrixIdentity();
	r.m[0][0] = c;  r.m[0][2] = -s;
	r.m[2][0] = s;  r.m[2][2] = c;
	return r;
}

Float4x4 MatrixRotationAxis(Float3 axis, float angle)
{
	Float3 n = Normalize(axis);
	float s = std::sin(angle), c = std::cos(angle), t = 1.0f - c;
	Float4x4 r = MatrixIdentity();
	r.m[0][0] = t * n.x * n.x + c;        r.m[0][1] = t * n.x * n.y + s * n.z;  r.m[0][2] = t * n.x * n.z - s * n.y;
	r.m[1][0] = t * n.x * n.y - s * n.z;  r.m[1][1] = t * n.y * n.y + c;        r.m[1][2] = t * n.y * n.z + s * n.x;
	r.m[2][0] = t * n.x * n.z + s * n.y;  r.m[2][1] = t * n.y * n.z - s * n.x;  r.m[2][2] = t * n.z * n.z + c;
	return r;
}

Float3 TransformNormal(Float3 v, const Float4x4& m)
{
	return { (v.z * m.m[2][0] + v.y * m.m[1][0]) + v.x * m.m[0][0],
	         (v.z * m.m[2][1] + v.y * m.m[1][1]) + v.x * m.m[0][1],
	         (v.z * m.m[2][2] + v.y * m.m[1][2]) + v.x * m.m[0][2] };
}

Camera::Camera()
	: mPosition{ 0, 0, 0 }, mRight{ 1, 0, 0 }, mUp{ 0, 1, 0 }, mLook{ 0, 0, 1 }
{
	mView = MatrixIdentity();
	SetLens(0.25f * 3.1415926535f, 1.0f, 1.0f, 2000.0f);
}

void Camera::SetPosition(float x, float y, float z) { mPosition = { x, y, z }; }

void Camera::SetLens(float fovY, float aspect, float zn, float zf)
{
	mFovY = fovY; mAspect = aspect; mNearZ = zn; mFarZ = zf;
	mProj = MatrixPerspectiveFovLH(fovY, aspect, zn, zf);
}

void Camera::Pitch(float angle)
{
	Float4x4 R = MatrixRotationAxis(mRight, angle);
	mUp = TransformNormal(mUp, R);
	mLook = TransformNormal(mLook, R);
}

void Camera::RotateY(float angle)
{
	Float4x4 R = MatrixRotationY(angle);
	mRight = TransformNormal(mRight, R);
	mUp = TransformNormal(mUp, R);
	mLook = TransformNormal(mLook, R);
}

void Camera::UpdateViewMatrix()
{
	Float3 L = Normalize(mLook);
	Float3 U = Normalize(Cross(L, mRight));
	Float3 R = Cross(U, L);
	float x = -Dot(mPosition, R), y = -Dot(mPosition, U), z = -Dot(mPosition, L);
	mRight = R; mUp = U; mLook = L;
	mView.m[0][0] = R.x; mView.m[1][0] = R.y; mView.m[2][0] = R.z; mView.m[3][0] = x;
	mView.m[0][1] = U.x; mView.m[1][1] = U.y; mView.m[2][1] = U.z; mView.m[3][1] = y;
	mView.m[0][2] = L.x; mView.m[1][2] = L.y; mView.m[2][2] = L.z; mView.m[3][2] = z;
	mView.m[0][3] = 0; mView.m[1][3] = 0; mView.m[2][3] = 0; mView.m[3][3] = 1;
}

} // namespace pth
