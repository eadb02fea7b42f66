import sys, os
sys.path.insert(0, "cuda-path-tracer_amd"); sys.path.insert(0, "tests")
from mipt import abi, device, host
import scenes
device.init(0)
big = host.HostScene.atrium(abi.SCENE_CRYTEK, 262000, 1024, 1)
cam, sun = host.camera_preset(abi.SCENE_CRYTEK, 1920, 1080, 1)
sc = device.Scene(big.desc)
img, ids, st = sc.render(cam, host.default_params(sun_pos=sun, global_illumination=1, num_samples=128, use_textures=1))
sn = st.node_visits - st.closest_node_visits; stt = st.triangle_tests - st.closest_triangle_tests
print("closest: rays %.0fM nodes/ray %.1f tris/ray %.2f" % (st.closest_rays/1e6, st.closest_node_visits/st.closest_rays, st.closest_triangle_tests/st.closest_rays))
print("shadow : rays %.0fM nodes/ray %.1f tris/ray %.2f" % (st.shadow_rays/1e6, sn/st.shadow_rays, stt/st.shadow_rays))
import numpy as np
print("lit fraction of primary hits:", float((img.sum(2) > 0).mean()))
