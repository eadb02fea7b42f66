/*
 * mipt_host.h - C ABI of the headless harness (host side only, no GPU code).
 *
 * The harness re-creates, without MFC / Direct3D, the callers that feed the render
 * call in the reference: scene assembly (CMainFrame::Initialize, MainFrm.cpp:228-492),
 * the camera and sun presets (CRayTracerView::PositionCameras / SetSunPos,
 * RayTracerView.cpp:221-238, 1043-1057), camera marshalling (CalcRayCUDA,
 * RayTracerView.cpp:794-806) and the 8-bit output stage (RayTracerView.cpp:925-931,
 * CPUPathTracer.cpp:509).  It produces the plain arrays of include/mipt.h.
 */
#ifndef MIPT_HOST_H
#define MIPT_HOST_H

#include <stdint.h>
#include "mipt.h"

#ifdef __cplusplus
extern "C" {
#endif

/* MeshType of framework.h:160-165 */
#define MIPT_SCENE_SPONZA   0
#define MIPT_SCENE_CRYTEK   1
#define MIPT_SCENE_CORNELL  2

typedef struct mipt_host_scene mipt_host_scene;

/* Loads an OBJ/MTL pair the way MainFrm.cpp:228-492 does: one mesh per `usemtl`
 * group, the first three vertices of each face, optional winding flip (Cornell box,
 * MainFrm.cpp:217,380-383), diffuse stored BGR, meshes ordered by triangle count
 * descending (a stable sort; the reference's std::sort is unstable).  Texture files
 * are decoded when they are PNG or baseline JPEG.  A material that names a diffuse texture which cannot
 * be loaded is left out of the material table, as in MainFrm.cpp:258-262: meshes that use
 * it then have no material (index -1).  Returns NULL on failure. */
mipt_host_scene* mipt_host_scene_load_obj(const char* path, int32_t flip_winding);

/* Deterministic procedural atrium used as a labelled stand-in for the Sponza meshes
 * that are absent from the reference tree.  scene_kind selects the coordinate scale of
 * the matching camera preset; texture_size is the edge of the procedural BGRA textures
 * and normal maps (0 = no textures). */
mipt_host_scene* mipt_host_scene_atrium(int32_t scene_kind, int64_t target_triangles,
                                        int32_t texture_size, uint32_t seed);
/* `copies` shifted instances of an existing scene flattened into one triangle list. */
mipt_host_scene* mipt_host_scene_instanced(const mipt_host_scene* base, int32_t copies, float spacing);

void mipt_host_scene_free(mipt_host_scene* s);
/* The descriptor stays valid until the scene is freed. */
const mipt_scene_desc* mipt_host_scene_desc(const mipt_host_scene* s);
const char* mipt_host_scene_label(const mipt_host_scene* s);

/* Camera of RayTracerView.cpp:211,221-234 (fovY 0.25*pi, near 1, far 6000, preset
 * position and RotateY) marshalled as CalcRayCUDA does (:794-806), and the sun
 * position of :235-237, 1043-1057. */
int mipt_host_camera_preset(int32_t scene_kind, int32_t client_width, int32_t client_height,
                            int32_t div, mipt_camera* cam, float sun_pos[3]);
/* General form: position, yaw about +y (Camera::RotateY) and pitch (Camera::Pitch). */
int mipt_host_camera_make(const float pos[3], float yaw, float pitch, float fov_y,
                          int32_t client_width, int32_t client_height, int32_t div, mipt_camera* cam);
/* Defaults of the reference's UI and CPU path: 16 samples, GI on, textures on,
 * normals off (SideView.cpp:81-119), intensity 3, epsilon 1e-20, seed 0x5EED. */
void mipt_host_default_params(mipt_params* p);

/* Output stage: float 255*rgb -> bytes by truncation.  BGRA DWORD as CPUPathTracer.cpp:509,
 * or 24-bit BGR rows as RayTracerView.cpp:925-931. */
void mipt_host_pack_bgra(const float* bgr255, int64_t num_pixels, uint32_t* out);
int mipt_host_write_ppm(const char* path, const float* bgr255, int32_t width, int32_t height);
/* The same bytes as a PNG file (8-bit RGB, no alpha); stands in for CImage::Save,
 * RayTracerView.cpp:934,1032. */
int mipt_host_write_png(const char* path, const float* bgr255, int32_t width, int32_t height);

/* Comparison of two images of the same size given as float 255*rgb arrays (num_values = 3 * pixels):
 * root mean square error, largest absolute difference, difference of the means, the standard error
 * of the mean of `b` taken over its values (sigma = std / sqrt(num_values)) and the share of values
 * that are identical.  A Monte Carlo image agrees with its reference when mean_diff <= 3 * sigma. */
typedef struct mipt_host_image_diff {
    double rmse, max_abs, mean_diff, sigma, identical;
} mipt_host_image_diff;
int mipt_host_compare_images(const float* a, const float* b, int64_t num_values, mipt_host_image_diff* out);
/* Reads an image file written by this library (binary PPM) or any PNG / baseline JPEG into float
 * 255*rgb values in BGR order; *out is malloc'ed, release with mipt_host_free. */
int mipt_host_read_image_bgr255(const char* path, float** out, int32_t* width, int32_t* height);

/* Decodes a PNG file to BGRA8 DWORD texels (byte0 = blue), the layout MainFrm.cpp:262-269
 * produces through CImage.  *out is malloc'ed; release with mipt_host_free. */
int mipt_host_load_png(const char* path, uint32_t** out, int32_t* width, int32_t* height);
/* The same for a PNG or a baseline JPEG file, told apart by their first bytes (CImage::Load takes
 * both, MainFrm.cpp:258; the default Sponza scene names JPEG textures).  Progressive JPEG files
 * are refused. */
int mipt_host_load_image(const char* path, uint32_t** out, int32_t* width, int32_t* height);
void mipt_host_free(void* p);

#ifdef __cplusplus
}
#endif
#endif /* MIPT_HOST_H */
