"""ctypes mirror of include/mipt.h and include/mipt_host.h (structures only)."""
import ctypes as C

MIPT_TEX_NONE, MIPT_TEX_F32X3, MIPT_TEX_BGRA8 = 0, 1, 2
SCENE_SPONZA, SCENE_CRYTEK, SCENE_CORNELL = 0, 1, 2


class Vertex(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("normal", C.c_float * 3), ("tex", C.c_float * 2)]


class Material(C.Structure):
    _fields_ = [("tex_data", C.c_void_p), ("tex_width", C.c_int32), ("tex_height", C.c_int32),
                ("tex_format", C.c_int32), ("nml_format", C.c_int32), ("nml_data", C.c_void_p),
                ("nml_width", C.c_int32), ("nml_height", C.c_int32), ("diffuse", C.c_float * 3),
                ("reserved", C.c_int32)]


class Mesh(C.Structure):
    _fields_ = [("first_triangle", C.c_int64), ("num_triangles", C.c_int64), ("material", C.c_int32),
                ("is_light", C.c_int32)]


class SceneDesc(C.Structure):
    _fields_ = [("vertices", C.POINTER(Vertex)), ("num_triangles", C.c_int64), ("meshes", C.POINTER(Mesh)),
                ("num_meshes", C.c_int32), ("num_materials", C.c_int32), ("materials", C.POINTER(Material))]


class Camera(C.Structure):
    _fields_ = [("client_width", C.c_int32), ("client_height", C.c_int32), ("div", C.c_int32),
                ("p0", C.c_float), ("p1", C.c_float), ("to_local", C.c_float * 16)]

    @property
    def image_size(self):
        if self.div <= 0:
            return 0, 0
        return self.client_width // self.div, self.client_height // self.div


class Params(C.Structure):
    _fields_ = [("num_samples", C.c_int32), ("sun_pos", C.c_float * 3), ("sun_intensity", C.c_float),
                ("global_illumination", C.c_int32), ("use_textures", C.c_int32), ("use_normals", C.c_int32),
                ("epsilon", C.c_float), ("normalize_normal_map", C.c_int32), ("seed", C.c_uint32),
                ("shard_rank", C.c_int32), ("shard_count", C.c_int32), ("tile_size", C.c_int32),
                ("sample_begin", C.c_int32), ("sample_end", C.c_int32), ("chunk_samples", C.c_int32),
                ("own_pixels_only", C.c_int32), ("count_box_tests", C.c_int32)]


class BuildOptions(C.Structure):
    _fields_ = [("sah_bins", C.c_int32), ("max_leaf_triangles", C.c_int32), ("traversal_cost", C.c_float),
                ("reserved", C.c_int32)]


class SceneInfo(C.Structure):
    _fields_ = [("num_triangles", C.c_int64), ("num_nodes", C.c_int64), ("num_leaves", C.c_int64),
                ("max_depth", C.c_int32), ("num_materials", C.c_int32), ("geometry_bytes", C.c_int64),
                ("texture_bytes", C.c_int64), ("build_ms", C.c_double), ("upload_ms", C.c_double),
                ("sah_cost", C.c_double), ("node_width", C.c_int32), ("reserved", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("closest_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("node_visits", C.c_uint64),
                ("triangle_tests", C.c_uint64), ("closest_node_visits", C.c_uint64),
                ("closest_triangle_tests", C.c_uint64), ("active_pixels", C.c_uint64), ("shadow_rays_culled", C.c_uint64),
                ("closest_ms", C.c_double), ("shadow_ms", C.c_double), ("shade_ms", C.c_double),
                ("total_ms", C.c_double), ("closest_launches", C.c_int32), ("shadow_launches", C.c_int32),
                ("chunk_samples", C.c_int32), ("samples_done", C.c_int32), ("closest_box_tests", C.c_uint64),
                ("shadow_box_tests", C.c_uint64), ("shaded_hits", C.c_uint64), ("bounce_rays_written", C.c_uint64),
                ("shade_bounce_ms", C.c_double), ("shade_bounce_launches", C.c_int32), ("reserved", C.c_int32)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_ if not n.startswith("reserved")}


PROGRESS_FN = C.CFUNCTYPE(None, C.c_int64, C.c_int64, C.c_void_p)

# every symbol include/mipt.h declares
E_INVALID, E_DEVICE, E_NOMEM, E_LIMIT, E_CANCELLED = -1, -2, -3, -4, -5     # MIPT_E_* of include/mipt.h
TRACE_DEVICE_MAX_RAYS = 1 << 18

MIPT_SYMBOLS = [
    "mipt_init", "mipt_last_error", "mipt_device_name", "mipt_scene_create", "mipt_scene_destroy",
    "mipt_scene_get_info", "mipt_scene_bvh_sizes", "mipt_scene_bvh_grid", "mipt_scene_download_bvh", "mipt_trace_rays",
    "mipt_trace_rays_device",
    "mipt_frame_create", "mipt_frame_destroy", "mipt_frame_set_params", "mipt_frame_accumulate",
    "mipt_frame_finalize", "mipt_frame_finalize_partial", "mipt_frame_cancel", "mipt_frame_samples_done",
    "mipt_frame_primary_ids", "mipt_frame_num_pixels", "mipt_render",
    "mipt_device_alloc", "mipt_device_free", "mipt_copy_to_host", "mipt_copy_to_device",
    "mipt_device_synchronize",
]
# every symbol include/mipt_host.h declares
class ImageDiff(C.Structure):
    _fields_ = [("rmse", C.c_double), ("max_abs", C.c_double), ("mean_diff", C.c_double), ("sigma", C.c_double),
                ("identical", C.c_double)]


MIPT_HOST_SYMBOLS = [
    "mipt_host_scene_load_obj", "mipt_host_scene_atrium", "mipt_host_scene_instanced", "mipt_host_scene_free",
    "mipt_host_scene_desc", "mipt_host_scene_label", "mipt_host_camera_preset", "mipt_host_camera_make",
    "mipt_host_default_params", "mipt_host_pack_bgra", "mipt_host_write_ppm", "mipt_host_write_png", "mipt_host_compare_images",
    "mipt_host_read_image_bgr255", "mipt_host_load_png", "mipt_host_load_image",
    "mipt_host_free",
]
