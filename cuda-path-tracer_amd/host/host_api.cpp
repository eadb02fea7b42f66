// C ABI of the headless harness (include/mipt_host.h): scene construction, camera and
// sun presets, parameter defaults, 8-bit output stage.
#include "../../include/mipt_host.h"
#include "Camera.h"
#include "Scene.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace pth;

struct mipt_host_scene { HostScene* scene; };

extern "C" {

mipt_host_scene* mipt_host_scene_load_obj(const char* path, int32_t flip_winding)
{
	if (!path) return nullptr;
	HostScene* s = AssembleObjScene(path, flip_winding != 0);
	if (!s) return nullptr;
	return new mipt_host_scene{ s };
}

mipt_host_scene* mipt_host_scene_atrium(int32_t scene_kind, int64_t target_triangles, int32_t texture_size, uint32_t seed)
{
	HostScene* s = BuildAtrium(scene_kind, target_triangles, texture_size, seed);
	return s ? new mipt_host_scene{ s } : nullptr;
}

mipt_host_scene* mipt_host_scene_instanced(const mipt_host_scene* base, int32_t copies, float spacing)
{
	if (!base || copies < 1) return nullptr;
	return new mipt_host_scene{ BuildInstanced(*base->scene, copies, spacing) };
}

void mipt_host_scene_free(mipt_host_scene* s)
{
	if (!s) return;
	delete s->scene;
	delete s;
}

const mipt_scene_desc* mipt_host_scene_desc(const mipt_host_scene* s) { return s ? s->scene->Desc() : nullptr; }
const char* mipt_host_scene_label(const mipt_host_scene* s) { return s ? s->scene->label.c_str() : ""; }

// CalcRayCUDA's camera marshalling, RayTracerView.cpp:794-806: P0 = Proj(0,0), P1 = Proj(1,1),
// toLocal = inverse(View) * inverse(Identity)
static int MarshalCamera(Camera& c, int32_t w, int32_t h, int32_t div, mipt_camera* cam)
{
	c.UpdateViewMatrix();
	Float4x4 invView, invWorld, world = MatrixIdentity();
	if (!MatrixInverse(c.View(), invView) || !MatrixInverse(world, invWorld)) return MIPT_E_INVALID;
	Float4x4 toLocal = MatrixMultiply(invView, invWorld);
	cam->client_width = w; cam->client_height = h; cam->div = div;
	cam->p0 = c.Proj()(0, 0); cam->p1 = c.Proj()(1, 1);
	memcpy(cam->to_local, toLocal.m, sizeof(cam->to_local));
	return MIPT_OK;
}

int mipt_host_camera_preset(int32_t scene_kind, int32_t client_width, int32_t client_height, int32_t div,
                            mipt_camera* cam, float sun_pos[3])
{
	if (!cam || client_width <= 0 || client_height <= 0 || div <= 0) return MIPT_E_INVALID;
	Camera c;
	c.Pitch(0);                                                        // RayTracerView.cpp:210
	c.SetLens(0.25f * 3.1415926535f, (float)client_width / client_height, 1.0f, 6000.0f); // :211, RayTracerView.h:112-113
	c.SetPosition(0, 0, 0);
	float dist = 2000;                                                 // :1046-1055
	if (scene_kind == MIPT_SCENE_CRYTEK) { c.SetPosition(-1280, 138, -50); c.RotateY((float)(3.1416 / 2)); dist = 2000; }   // :223-226
	else if (scene_kind == MIPT_SCENE_SPONZA) { c.SetPosition(10, 5, 0); c.RotateY((float)(-2 * 3.1416 / 4)); dist = 80; }   // :227-230
	else if (scene_kind == MIPT_SCENE_CORNELL) { c.SetPosition(0, 1.5, 4); c.RotateY((float)3.1416); dist = 20; }            // :231-234
	else return MIPT_E_INVALID;
	if (sun_pos) {
		float nPos = 900.0f / 2000.0f;                                 // :235-237
		float nAngle = (float)(3.1415926535897932384626433832795 * nPos);  // M_PI is a double there (:34)
		sun_pos[0] = 0; sun_pos[1] = dist * std::sin(nAngle); sun_pos[2] = dist * std::cos(nAngle);   // :1057
	}
	return MarshalCamera(c, client_width, client_height, div, cam);
}

int mipt_host_camera_make(const float pos[3], float yaw, float pitch, float fov_y, int32_t client_width,
                          int32_t client_height, int32_t div, mipt_camera* cam)
{
	if (!cam || !pos || client_width <= 0 || client_height <= 0 || div <= 0) return MIPT_E_INVALID;
	Camera c;
	c.SetLens(fov_y, (float)client_width / client_height, 1.0f, 6000.0f);
	c.SetPosition(pos[0], pos[1], pos[2]);
	c.RotateY(yaw);
	c.Pitch(pitch);
	return MarshalCamera(c, client_width, client_height, div, cam);
}

void mipt_host_default_params(mipt_params* p)
{
	if (!p) return;
	memset(p, 0, sizeof(*p));
	p->num_samples = 16;              // SideView.cpp:85
	p->sun_intensity = 3.0f;          // framework.h:99
	p->global_illumination = 1;
	p->use_textures = 1;
	p->use_normals = 0;
	p->epsilon = 1e-20f;              // xnacollision.cpp:1073-1076
	p->normalize_normal_map = 0;
	p->seed = 0x5EED;
	p->shard_rank = 0; p->shard_count = 1; p->tile_size = 32;
}

void mipt_host_pack_bgra(const float* bgr255, int64_t num_pixels, uint32_t* out)
{
	for (int64_t i = 0; i < num_pixels; i++) {
		uint32_t b = (uint32_t)(uint64_t)bgr255[3 * i] & 255u, g = (uint32_t)(uint64_t)bgr255[3 * i + 1] & 255u,
		         r = (uint32_t)(uint64_t)bgr255[3 * i + 2] & 255u;
		out[i] = b | (g << 8) | (r << 16) | (255u << 24);
	}
}

int mipt_host_write_ppm(const char* path, const float* bgr255, int32_t width, int32_t height)
{
	FILE* f = fopen(path, "wb");
	if (!f) return MIPT_E_INVALID;
	fprintf(f, "P6\n%d %d\n255\n", width, height);
	std::vector<unsigned char> row((size_t)width * 3);
	for (int y = 0; y < height; y++) {
		for (int x = 0; x < width; x++) {
			const float* p = &bgr255[3 * ((size_t)y * width + x)];
			row[3 * x + 0] = (unsigned char)p[2];     // byte truncation as RayTracerView.cpp:927-929
			row[3 * x + 1] = (unsigned char)p[1];
			row[3 * x + 2] = (unsigned char)p[0];
		}
		fwrite(row.data(), 1, row.size(), f);
	}
	fclose(f);
	return MIPT_OK;
}

int mipt_host_write_png(const char* path, const float* bgr255, int32_t width, int32_t height)
{
	if (!path || !bgr255 || width <= 0 || height <= 0) return MIPT_E_INVALID;
	std::vector<uint32_t> px((size_t)width * height);
	mipt_host_pack_bgra(bgr255, (int64_t)px.size(), px.data());
	return WritePng(path, px, width, height) ? MIPT_OK : MIPT_E_INVALID;
}

int mipt_host_compare_images(const float* a, const float* b, int64_t n, mipt_host_image_diff* out)
{
	if (!a || !b || !out || n <= 0) return MIPT_E_INVALID;
	double se = 0, mx = 0, sa = 0, sb = 0, sbb = 0; int64_t same = 0;
	for (int64_t i = 0; i < n; i++) {
		double d = (double)a[i] - (double)b[i];
		se += d * d; if (fabs(d) > mx) mx = fabs(d);
		sa += a[i]; sb += b[i]; sbb += (double)b[i] * b[i];
		same += a[i] == b[i];
	}
	double mb = sb / n, var = sbb / n - mb * mb;
	out->rmse = sqrt(se / n); out->max_abs = mx; out->mean_diff = fabs(sa / n - mb);
	out->sigma = sqrt((var > 0 ? var : 0) / n); out->identical = (double)same / n;
	return MIPT_OK;
}

int mipt_host_read_image_bgr255(const char* path, float** out, int32_t* width, int32_t* height)
{
	if (!path || !out || !width || !height) return MIPT_E_INVALID;
	*out = nullptr;
	std::vector<uint32_t> px; int32_t w = 0, h = 0;
	FILE* f = fopen(path, "rb");
	if (!f) return MIPT_E_INVALID;
	int maxv = 0;
	bool ppm = fscanf(f, "P6 %d %d %d", &w, &h, &maxv) == 3 && maxv == 255 && w > 0 && h > 0 && fgetc(f) != EOF;
	if (ppm) {
		std::vector<unsigned char> rgb((size_t)w * h * 3);
		ppm = fread(rgb.data(), 1, rgb.size(), f) == rgb.size();
		px.resize((size_t)w * h);
		for (size_t i = 0; ppm && i < px.size(); i++) px[i] = (uint32_t)rgb[3 * i + 2] | ((uint32_t)rgb[3 * i + 1] << 8) | ((uint32_t)rgb[3 * i] << 16) | 0xff000000u;
	}
	fclose(f);
	if (!ppm && !LoadImageFile(path, px, w, h)) return MIPT_E_INVALID;
	float* p = (float*)malloc(sizeof(float) * 3 * px.size());
	if (!p) return MIPT_E_NOMEM;
	for (size_t i = 0; i < px.size(); i++) { p[3 * i] = (float)(px[i] & 255u); p[3 * i + 1] = (float)((px[i] >> 8) & 255u); p[3 * i + 2] = (float)((px[i] >> 16) & 255u); }
	*out = p; *width = w; *height = h;
	return MIPT_OK;
}

int mipt_host_load_png(const char* path, uint32_t** out, int32_t* width, int32_t* height)
{
	std::vector<uint32_t> px; int32_t w = 0, h = 0;
	if (!path || !out || !LoadPng(path, px, w, h)) return MIPT_E_INVALID;
	*out = (uint32_t*)malloc(px.size() * 4);
	if (!*out) return MIPT_E_NOMEM;
	memcpy(*out, px.data(), px.size() * 4);
	if (width) *width = w;
	if (height) *height = h;
	return MIPT_OK;
}

int mipt_host_load_image(const char* path, uint32_t** out, int32_t* width, int32_t* height)
{
	std::vector<uint32_t> px; int32_t w = 0, h = 0;
	if (!path || !out || !LoadImageFile(path, px, w, h)) return MIPT_E_INVALID;
	*out = (uint32_t*)malloc(px.size() * 4);
	if (!*out) return MIPT_E_NOMEM;
	memcpy(*out, px.data(), px.size() * 4);
	if (width) *width = w;
	if (height) *height = h;
	return MIPT_OK;
}

void mipt_host_free(void* p) { free(p); }

} // extern "C"
