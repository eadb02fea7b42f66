"""Headline benchmark: Mrays/s of the wavefront path tracer on the Crytek-Sponza workload
(direct + GI, 128 first-bounce samples, 1920x1080), one process per GPU.

  python bench.py --gpus 1 --steps 3 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W

A step is one whole frame: every rank traces its image tiles (round-robin 32x32 tiles, all 128
samples), the per-pixel first-bounce sums are reduced to rank 0 over RCCL and rank 0 applies the
final clamp.  The Sponza meshes are not part of the reference tree, so the scene is the
harness's procedural atrium with the same triangle count, material count and texture sizes
(labelled synthetic in the output).  Rays are counted on the device (closest-hit + shadow
queries actually issued).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "cuda-path-tracer_amd"))

WORKLOADS = {
    # name: (scene kind, triangles, texture edge, samples, gi, width, height, div)
    "crytek_gi128_1080p": ("crytek", 262000, 1024, 128, 1, 1920, 1080, 1),
    "sponza_gi128_1080p": ("sponza", 66000, 1024, 128, 1, 1920, 1080, 1),
    "sponza_direct_1080p": ("sponza", 66000, 1024, 64, 0, 1920, 1080, 1),
    "crytek_gi16_270p": ("crytek", 262000, 256, 16, 1, 1920, 1080, 4),
    # 38 shifted instances flattened into one ~10 M-triangle list, 3840x2160, 512 samples
    "crytek38x_gi512_2160p": ("crytek", 262000, 1024, 512, 1, 3840, 2160, 1, 38),
}

FP32_VECTOR_PEAK_TFLOPS = 157.3     # plain v_fma_f32: 256 CUs x 4 SIMDs x 32 lanes per cycle x 2 flop x 2.4 GHz
HBM_PEAK_GBS = 8000.0
FLOP_PER_TRI_TEST = 51              # full Moeller-Trumbore from three vertices
FLOP_PER_BOX_TEST = 22
BYTES_PER_NODE_CHILD = 16          # a node slot holds, per child, a 12-byte box on the 16-bit grid and a 4-byte link
BYTES_PER_TRI_TEST = 36             # three float3 positions


def cpu_baseline(scene, cam, params, budget_s=15.0):
    """The CPU oracle (the restated reference CPU path with its octree) on this host's cores, on a
    bounded sample of image rows of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    osc = ob.OracleScene(scene.desc)
    w, h = cam.image_size
    # the GPU box gives a job 16 host cores of a much larger machine (os.cpu_count() reports the machine)
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    # calibrate on a few rows, then size the sample to the time budget
    probe_step = max(1, h // 4)
    _, _, _, st = osc.render(cam, params, accel=ob.ACCEL_OCTREE, row_step=probe_step, row_phase=probe_step // 2, threads=threads)
    rows_probe = len(range(probe_step // 2, h, probe_step))
    per_row = max(st.seconds / rows_probe, 1e-6)
    rows = int(max(rows_probe, min(h, budget_s / per_row)))
    step = max(1, h // rows)
    _, _, _, st = osc.render(cam, params, accel=ob.ACCEL_OCTREE, row_step=step, row_phase=step // 2, threads=threads)
    rows_done = len(range(step // 2, h, step))
    rays = st.closest_rays + st.shadow_rays
    out = {
        "value": rays / st.seconds / 1e6, "unit": "Mrays/s", "cores": st.threads, "host_cpus": os.cpu_count(), "kind": "port",
        "sample": f"{rows_done} of {h} image rows (every {step}th), {rays} rays in {st.seconds:.1f} s, octree search, OpenMP",
    }
    # the reference's octree (Mesh.cpp:88-186) drops triangles that lie on a face of a re-fitted box: the search
    # this baseline times sees only part of the scene (and does less work than a complete one would)
    total_tris, reachable_tris = osc.octree_reachable()
    out["octree_triangles_reachable"] = int(reachable_tris)
    out["octree_triangles_total"] = int(total_tris)
    out["sample"] += f"; the reference's octree holds {reachable_tris} of the {total_tris} triangles in a leaf, the others are never tested"
    # the reference's CPU path is serial (one thread renders the image): the same restatement on one core,
    # on the middle row of the image only
    step1 = h
    _, _, _, s1 = osc.render(cam, params, accel=ob.ACCEL_OCTREE, row_step=step1, row_phase=step1 // 2, threads=1)
    out["single_thread_value"] = (s1.closest_rays + s1.shadow_rays) / s1.seconds / 1e6
    out["single_thread_sample"] = f"{len(range(step1 // 2, h, step1))} rows, {s1.seconds:.1f} s"
    return out


def kernel_source_hash():
    """SHA-1 over the HIP sources the kernels are built from, in a fixed order: counter measurements are
    only quoted for the code they were taken on (profiles/pmc_passes.sh stores the same hash)."""
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(ROOT, "cuda-path-tracer_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()


def measured_traffic(workload, kernel="k_trace<0>"):
    """Fabric-side bytes per launch of one kernel from the committed counter run (profiles/traffic.json,
    written by profiles/pmc_report.py from separate FETCH_SIZE and WRITE_SIZE passes of this same
    command).  None when there is no entry for the workload and kernel, or when the entry was measured
    on other kernel sources than the ones in the tree."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f)
        e = t.get(workload, {}).get(kernel)
        if not e or e.get("source_sha1") != kernel_source_hash():
            return None
        return float(e["bytes_per_launch"])
    except (OSError, ValueError, KeyError, AttributeError):
        return None


def traversal_roofs(tflops, traffic, launch_s):
    """The two roofs of a traversal kernel and the one that names its bound: the larger share.
    valu: algorithmic flop (box and triangle tests) over the kernel time against the FP32 vector peak.
    hbm: the MEASURED fabric-side bytes of a launch (profiles/traffic.json, Infinity-Cache hits included) over its
    duration against the HBM peak.  (The bytes the lanes ask for, 64 per node visit and 36 per triangle test, are
    served by LDS, L1 and L2 at a rate above the HBM peak, so they say nothing about memory: requested_gbs.)
    Without a measured traffic figure for the tree's kernel sources only the first roof is known."""
    valu = {"achieved": tflops, "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / FP32_VECTOR_PEAK_TFLOPS}
    hbm = None
    if traffic and launch_s > 0:
        gbs = traffic / launch_s / 1e9
        hbm = {"achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS}
    first = ("hbm", hbm) if hbm and hbm["frac"] > valu["frac"] else ("valu", valu)
    return {"bound": first[0], **first[1], "valu": valu, "hbm": hbm}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="crytek_gi128_1080p", choices=sorted(WORKLOADS))
    ap.add_argument("--tris", type=int, default=0)
    ap.add_argument("--samples", type=int, default=0)
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--shard", default="tiles", choices=["tiles", "samples"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dump-image", default="", help="rank 0 saves the final image (float32, H x W x 3, BGR, 0..255) as .npy")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from mipt import abi, device, host, sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # MIPT_BENCH_BACKEND=gloo rehearses the multi-rank flow where the ranks have to share GPUs (the
    # sums then travel through host memory); the real run uses one GPU per rank and RCCL
    backend = os.environ.get("MIPT_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)              # before the communicator is created: it binds to the current device
    # MIPT_BENCH_FORCE_PG=1: a one-rank run also creates its process group and sends its accumulator through the
    # collective, so that the RCCL path (communicator bound to the device, collective stream -> library stream) is
    # exercised on a single GPU
    use_pg = world > 1 or os.environ.get("MIPT_BENCH_FORCE_PG") == "1"
    if use_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    device.init(local_rank)
    coll = "cuda" if backend == "nccl" else "cpu"      # where tensors handed to collectives live

    kind_name, tris, tex, samples, gi, width, height, div = WORKLOADS[args.workload][:8]
    copies = WORKLOADS[args.workload][8] if len(WORKLOADS[args.workload]) > 8 else 1
    tris = args.tris or tris
    samples = args.samples or samples
    kind = {"crytek": abi.SCENE_CRYTEK, "sponza": abi.SCENE_SPONZA}[kind_name]
    scene = host.HostScene.atrium(kind, tris, tex, 1)
    if copies > 1:
        scene = scene.instanced(copies, 4000.0 if kind == abi.SCENE_CRYTEK else 40.0)
    cam, sun = host.camera_preset(kind, width, height, div)
    dev = device.Scene(scene.desc, sah_bins=int(os.environ.get("MIPT_BENCH_BINS", "0")),
                       max_leaf_triangles=int(os.environ.get("MIPT_BENCH_LEAF", "0")),
                       traversal_cost=float(os.environ.get("MIPT_BENCH_TCOST", "0")))
    info = dev.info()

    # tile shards: every rank traces, accumulates and finalizes its own tiles and the final images are added
    # (disjoint pixels: exact); sample shards: the unclamped sums are added and rank 0 finalizes
    own_tiles = (world > 1 or os.environ.get("MIPT_BENCH_FORCE_PG") == "1") and args.shard == "tiles"
    shard = sharding.shard_fields(rank, world, samples, args.shard, own_pixels_only=own_tiles)
    # MIPT_BENCH_AS_SHARD=r/N (single process only): time the work of rank r of an N-rank run alone, to see how
    # the per-rank frame time falls with N where only one GPU is at hand
    if world == 1 and os.environ.get("MIPT_BENCH_AS_SHARD"):
        r_, n_ = (int(x) for x in os.environ["MIPT_BENCH_AS_SHARD"].split("/"))
        own_tiles = args.shard == "tiles"
        shard = sharding.shard_fields(r_, n_, samples, args.shard, own_pixels_only=own_tiles)
    params = host.default_params(sun_pos=sun, global_illumination=gi, num_samples=samples, use_textures=1, use_normals=0,
                                 chunk_samples=args.chunk, **shard)
    frame = device.Frame(dev, cam, params)
    npx = frame.num_pixels
    accum = torch.zeros(npx * 3, dtype=torch.float32, device="cuda")
    out = torch.zeros(npx * 3, dtype=torch.float32, device="cuda")

    def reduce(t):
        if use_pg and coll == "cpu":
            host_sum = t.cpu()
            sharding.reduce_to_rank0(host_sum, dist, world, force=use_pg)
            if rank == 0:
                t.copy_(host_sum)
        else:
            sharding.reduce_to_rank0(t, dist, world, force=use_pg)

    def step():
        st = frame.accumulate(accum.data_ptr())
        if own_tiles:
            frame.finalize(accum.data_ptr(), out.data_ptr())
            reduce(out)
        else:
            reduce(accum)
            if rank == 0:
                frame.finalize(accum.data_ptr(), out.data_ptr())
        return st

    def fence():
        if use_pg:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    stats = [step() for _ in range(args.steps)]
    fence()
    elapsed = time.perf_counter() - t0

    # one more frame, untimed, with the closest-hit kernel counting the child boxes that exist in the nodes it
    # visits (the frames are deterministic: the count holds for the timed ones): a node visit tests 2..4 boxes
    params.count_box_tests = 1
    frame.set_params(cam, params)
    counted = frame.accumulate(accum.data_ptr())
    boxes_per_visit = counted.closest_box_tests / max(counted.closest_node_visits, 1)
    shadow_boxes_per_visit = counted.shadow_box_tests / max(counted.node_visits - counted.closest_node_visits, 1)

    tot = torch.tensor([
        float(sum(s.closest_rays + s.shadow_rays for s in stats)),
        float(sum(s.closest_triangle_tests for s in stats)), float(sum(s.closest_node_visits for s in stats)),
        float(sum(s.closest_ms for s in stats)), float(sum(s.closest_launches for s in stats)),
        float(sum(s.triangle_tests for s in stats)), float(sum(s.node_visits for s in stats)),
        float(sum(s.shadow_ms for s in stats)), float(sum(s.closest_rays for s in stats)),
        float(sum(s.shade_ms for s in stats)), float(sum(s.shadow_rays_culled for s in stats)),
        float(sum(s.shade_bounce_ms for s in stats)), float(sum(s.shade_bounce_launches for s in stats)),
        float(sum(s.shaded_hits for s in stats)), float(sum(s.bounce_rays_written for s in stats)),
        float(sum(s.shadow_launches for s in stats)), float(sum(s.shadow_rays for s in stats)),
    ], dtype=torch.float64, device=coll)
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=coll)
    if use_pg:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        # rays and counters add over ranks; kernel time is per-rank wall time of rank 0's kernels
        sums = tot.clone()
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        total_rays, culled_rays = sums[0].item(), sums[10].item()
    else:
        total_rays, culled_rays = tot[0].item(), tot[10].item()
    elapsed = tmax.item()

    if rank == 0:
        import hashlib
        final = out.cpu().numpy()
        image_sha1 = hashlib.sha1(final.tobytes()).hexdigest()   # tile shards add up bit for bit: same hash at any N
        if args.dump_image:
            import numpy as np
            np.save(args.dump_image, final.reshape(cam.image_size[1], cam.image_size[0], 3))
        s0 = stats[-1]
        closest_s = tot[3].item() / 1e3                 # rank 0's closest-hit kernel time over the timed steps
        width = int(info.node_width)                    # child slots per node
        # algorithmic work of the closest-hit kernel: box tests against children that exist, triangle tests
        flops = FLOP_PER_TRI_TEST * tot[1].item() + FLOP_PER_BOX_TEST * boxes_per_visit * tot[2].item()
        bytes_ = BYTES_PER_TRI_TEST * tot[1].item() + BYTES_PER_NODE_CHILD * width * tot[2].item()
        achieved_tf = flops / closest_s / 1e12 if closest_s > 0 else 0.0
        launches = max(tot[4].item(), 1.0)
        traffic = measured_traffic(args.workload)
        valu_frac = achieved_tf / FP32_VECTOR_PEAK_TFLOPS
        hbm_frac = (traffic / (closest_s / launches) / 1e9) / HBM_PEAK_GBS if traffic and closest_s > 0 else None
        # any-hit traversal: the same units of work as the closest-hit kernel
        shadow_s = tot[7].item() / 1e3
        sh_tests, sh_visits = tot[5].item() - tot[1].item(), tot[6].item() - tot[2].item()
        sh_flops = FLOP_PER_TRI_TEST * sh_tests + FLOP_PER_BOX_TEST * shadow_boxes_per_visit * sh_visits
        sh_tf = sh_flops / shadow_s / 1e12 if shadow_s > 0 else 0.0
        sh_launches = max(tot[15].item(), 1.0)
        sh_traffic = measured_traffic(args.workload, "k_trace<1>")
        # bounce shading: per shaded hit one 16-byte hit record, one 128-byte shading record and one 4-byte texel are
        # read and a 16-byte path state is written; per shadow ray and per ray of the next depth 32 bytes are written;
        # a ray that missed costs its hit record and a zero path state
        bounce_s = tot[11].item() / 1e3
        b_launches = max(tot[12].item(), 1.0)
        hits, next_rays = tot[13].item(), tot[14].item()
        bounce_in = tot[8].item() - args.steps * npx if world == 1 else None          # rays of depth 1..3 (one rank: all but the primary rays)
        shadow_bounce = tot[16].item()                     # includes the primary level's shadow rays (at most one per pixel)
        b_bytes = hits * (16 + 128 + 4 + 16) + 32 * shadow_bounce + 32 * next_rays + ((bounce_in - hits) * 32 if bounce_in else 0)
        b_gbs = b_bytes / bounce_s / 1e9 if bounce_s > 0 else 0.0
        b_traffic = measured_traffic(args.workload, "k_shade_bounce")
        result = {
            "metric": "Mrays/s (1 & 8 GPU) + render time, Crytek Sponza GI 128 spp 1080p",
            "value": total_rays / elapsed / 1e6, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": args.workload, "scene": scene.label, "triangles": int(info.num_triangles),
                "materials": int(info.num_materials), "image": f"{cam.image_size[0]}x{cam.image_size[1]}",
                "samples": samples, "global_illumination": bool(gi), "textures": True, "normal_maps": False,
                "parallelism": f"{args.shard}x{world}", "rays_per_frame": total_rays / args.steps,
                # `value` counts the rays that were traversed.  The reference casts a shadow ray at every hit; the
                # ones whose answer cannot change the image (direct term +0) are not traversed here.  With them the
                # frame has reference_rays_per_frame rays, the number an implementation that traces all of them reports.
                "shadow_rays_not_traversed_per_frame": culled_rays / args.steps,
                "reference_rays_per_frame": (total_rays + culled_rays) / args.steps,
                "reference_mrays_per_s": (total_rays + culled_rays) / elapsed / 1e6,
                "render_time_s": elapsed / args.steps, "bvh_build_ms": info.build_ms, "bvh_depth": int(info.max_depth),
                "bvh_sah_cost": info.sah_cost, "bvh_node_width": width, "image_sha1": image_sha1, "samples_per_pass": int(s0.chunk_samples),
            },
            "roofline": {
                "kernel": "k_trace<0> (closest-hit BVH traversal)",
                **traversal_roofs(achieved_tf, traffic, closest_s / launches),
                "traffic": traffic,
                "traffic_unit": "bytes per launch, fabric side of L2 (profiles/traffic.json)",
                "algorithmic_bytes_per_launch": bytes_ / launches,
                "avg_launch_ms": 1e3 * closest_s / launches, "launches": int(launches),
                "triangle_tests_per_s": tot[1].item() / closest_s if closest_s > 0 else 0.0,
                "node_visits_per_s": tot[2].item() / closest_s if closest_s > 0 else 0.0,
                "closest_rays_per_s": tot[8].item() / closest_s if closest_s > 0 else 0.0,
                # bytes the lanes ask the caches for (whole node slots and triangle positions); most of them are
                # served by LDS, L1 and L2: this is a request rate, not memory traffic
                "requested_gbs": bytes_ / closest_s / 1e9 if closest_s > 0 else 0.0, "hbm_peak_gbs": HBM_PEAK_GBS,
                # share of the HBM peak that the measured fabric-side traffic of a launch amounts to
                "hbm_frac": hbm_frac,
                "box_tests_per_node_visit": boxes_per_visit,
                "closest_ms_per_step": [round(s.closest_ms, 3) for s in stats], "shadow_ms_per_step": [round(s.shadow_ms, 3) for s in stats],
                "kernel_time_share": {"closest": tot[3].item() / (1e3 * elapsed), "shadow": tot[7].item() / (1e3 * elapsed),
                                      "shading": tot[9].item() / (1e3 * elapsed)},
            },
        }
        result["roofline_kernels"] = [
            {"kernel": "k_trace<1> (any-hit BVH traversal of the shadow rays)",
             **traversal_roofs(sh_tf, sh_traffic, shadow_s / sh_launches),
             "traffic": sh_traffic, "avg_launch_ms": 1e3 * shadow_s / sh_launches, "launches": int(sh_launches),
             "ms_per_step": 1e3 * shadow_s / args.steps, "box_tests_per_node_visit": shadow_boxes_per_visit,
             "triangle_tests_per_s": sh_tests / shadow_s if shadow_s > 0 else 0.0, "node_visits_per_s": sh_visits / shadow_s if shadow_s > 0 else 0.0,
             "hbm_frac": (sh_traffic / (shadow_s / sh_launches) / 1e9) / HBM_PEAK_GBS if sh_traffic and shadow_s > 0 else None},
            {"kernel": "k_shade_bounce (surface, light and next sample at depth 1..3)", "bound": "hbm", "achieved": b_gbs,
             "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": b_gbs / HBM_PEAK_GBS, "traffic": b_traffic,
             "algorithmic_bytes_per_launch": b_bytes / b_launches, "avg_launch_ms": 1e3 * bounce_s / b_launches, "launches": int(b_launches),
             "ms_per_step": 1e3 * bounce_s / args.steps, "shaded_hits_per_s": hits / bounce_s if bounce_s > 0 else 0.0,
             # measured fabric-side traffic (Infinity Cache hits included) over the launch time, against the HBM peak
             "traffic_frac": (b_traffic / (bounce_s / b_launches) / 1e9) / HBM_PEAK_GBS if b_traffic and bounce_s > 0 else None},
        ]
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(scene, cam, params)
            result["cpu_baseline"]["gpu_over_cpu"] = result["value"] / max(result["cpu_baseline"]["value"], 1e-9)
        if use_pg:
            result["config"]["process_group"] = f"{backend}x{world}"
        print(json.dumps(result))
    if use_pg:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
