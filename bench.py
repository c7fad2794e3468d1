#!/usr/bin/env python3
"""Benchmark of the hot path: Mrays/s on the Cornell box at 1920x1080, and the time to 1024 samples per pixel.

One "step" renders SPP_PER_STEP samples of every pixel on each GPU (generate / extend / shade / accumulate,
all bounces) and — when more than one GPU takes part — adds the step's radiance accumulators with one RCCL
all-reduce.  Inputs (scene tables, camera) are resident in HBM before the timed region; the accumulators stay
on the device.  One ray = one closest-hit query (`world.hit`, src/renderer.rs:44 of the reference).

    python bench.py [--gpus N] [--steps K] [--warmup W]
N > 1 runs one rank per GPU: either a launcher (torch.distributed.run) has started the N ranks, or bench.py
starts them itself as a child process.

After the timed steps the run renders exactly `--total-spp` (1024) samples per pixel split over the ranks, with one
exchange and one download, and reports that wall time next to the figure extrapolated from the steps' rate.

At N = 1 the same run also times the other configurations (teapot, 485 spheres, 10 M triangle soup: `legs` in
the JSON line), each with its own roofline, and the CPU oracle on the host cores (`cpu_baseline`).

    python bench.py --profile-leg NAME      three launches of that leg's reference workload (samples [0, spp)),
                                            the program rocprofv3 is pointed at (tools/prof_legs.sh)
"""
from __future__ import annotations

import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

WIDTH, HEIGHT, MAX_DEPTH, SEED = 1920, 1080, 50, 1  # --width / --height override the size for rehearsals only
SPP_PER_STEP = 64
MAX_PATHS = 1 << 27          # path slots per batch: the 64 samples of a 1080p frame a step renders
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E peak
SIMDS, CLOCK_HZ = 1024, 2.4e9  # 256 CUs x 4 SIMDs; the chip's maximum shader clock
# wave64 VALU instructions per second: a SIMD is 32 lanes wide, so one instruction every two cycles (the chip's
# 157.3 TFLOP/s of vector binary32 is this rate x 64 lanes x 2 flops of a fused multiply-add)
VALU_PEAK = SIMDS * CLOCK_HZ / 2.0
FP32_PEAK_FLOPS = VALU_PEAK * 64 * 2  # 157.3 TFLOP/s of vector binary32
# Algorithmic flops per test: a box test is 6 subtractions, 6 multiplications, 3 divisions (the reference recomputes
# 1 / d per node) and 15 compares / selects = 30; a Moeller-Trumbore triangle test 45; a sphere test 28 (three dot
# products, the discriminant, one square root, two divisions, the range compares).
FLOPS_PER_BOX, FLOPS_PER_TRIANGLE, FLOPS_PER_SPHERE = 30, 45, 28
# What tools/calib/valu_issue.hip sustains on the chip with 6 waves per SIMD (profiles/r02_valu_issue.txt): streams of
# independent add/mul/fma/mov/and/xor reach 0.70-0.80 of VALU_PEAK (the clock falls to 2.05-2.2 GHz under them);
# min/max/compare/select/shift/integer multiply/convert/div_fixup issue once per 4.2 cycles, rcp/sqrt once per 8.2.
VALU_SUSTAINED = {"full_rate_ops": 8.6e11, "half_rate_ops": 5.7e11, "quarter_rate_ops": 3.0e11}
PIPELINES = {1: "split", 2: "fused", 3: "fused+lds", 4: "hybrid", 5: "hybrid+lds", 6: "split+lds", 7: "paths",
             8: "paths+lds", 9: "lanes"}

# The configurations of the metric: (spp per launch, background).  Every launch of a leg renders the same samples
# [0, spp), so launches are identical and comparable with the profiled ones.
LEGS = {
    "cornell": (SPP_PER_STEP, (0.0, 0.0, 0.0)),
    "teapot": (SPP_PER_STEP, (0.0, 0.0, 0.0)),        # the reference's asset as it would run it: default camera
    "teapot_lit": (SPP_PER_STEP, (0.0, 0.0, 0.0)),    # the same mesh with a light and a camera outside it
    "spheres": (SPP_PER_STEP, (0.7, 0.8, 1.0)),
    "soup10m": (64, (0.0, 0.0, 0.0)),
}


def source_digest():
    """Digest of the kernel sources: a committed PMC profile only speaks for the build it was taken from."""
    h = hashlib.sha256()
    for name in ("kernels.hip", "rt_math.h", "rt_device.h", "kernels.h"):
        h.update(open(os.path.join(ROOT, "raytracer_amd", "csrc", name), "rb").read())
    return h.hexdigest()[:16]


def leg_scene(name):
    from raytracer_amd import load_from_gltf
    from raytracer_amd.scenes import lit_teapot, rtiow_final_scene, triangle_soup

    if name == "cornell":
        desc = load_from_gltf(os.path.join(ROOT, "assets", "cornell.gltf"))
        return desc, desc.camera(0), "assets/cornell.gltf Cornell box"
    if name in ("teapot", "teapot_lit"):
        desc = load_from_gltf(os.path.join(ROOT, "assets", "teapot.gltf"))
        if name == "teapot":
            return desc, desc.camera(0), ("assets/teapot.gltf, 9128 triangles, default camera (no light: black image, "
                                          "every path traced)")
        desc, cs = lit_teapot(desc)
        return desc, cs, "assets/teapot.gltf + a 30x30 light, camera at (0, 10, 60)"
    if name == "spheres":
        desc, cs = rtiow_final_scene()
        return desc, cs, ("RTIOW final scene, 485 spheres (lambertian / metal / dielectric), defocus, "
                          "background (0.7, 0.8, 1.0)")
    if name == "soup10m":
        desc, cs = triangle_soup(10_000_000)
        return desc, cs, "synthetic soup of 10 000 000 triangles (0.1 % emitters), incoherent diffuse bounces"
    raise SystemExit(f"unknown leg {name}")


def committed_profile(name, kernel_hint, rays_per_launch):
    """The PMC figures of profiles/r03_pmc_<leg>.json, or (None, reason) when they do not describe this build and
    this workload: the kernel sources must hash the same and the profiled launch must have traced the same rays."""
    path = os.path.join(ROOT, "profiles", f"r03_pmc_{name}.json")
    if not os.path.exists(path):
        return None, "no committed PMC profile"
    p = json.load(open(path))
    if p.get("source_digest") != source_digest():
        return None, (f"stale PMC profile: taken from kernel sources {p.get('source_digest')}, this build is "
                      f"{source_digest()}")
    if int(p.get("rays_per_dispatch", -1)) != int(rays_per_launch):
        return None, f"stale PMC profile: {p.get('rays_per_dispatch')} rays per dispatch there, {rays_per_launch} here"
    if kernel_hint not in p.get("kernel", ""):
        return None, f"stale PMC profile: kernel {p.get('kernel')!r}, expected {kernel_hint}"
    return p, os.path.relpath(path, ROOT)


def committed_kernel_stats(name, kernels):
    """Average duration (ms) of the render kernels `kernels` ("a + b" when a launch is two kernels) in the committed
    `rocprofv3 --kernel-trace --stats` summary of this leg: of the default bench command for Cornell, of the leg's
    own launches otherwise.  (None, None) without a summary or when a kernel is not in it."""
    import csv

    path = os.path.join(ROOT, "profiles", f"r03_{'bench' if name == 'cornell' else name}_kernel_stats.csv")
    if not os.path.exists(path):
        return None, None
    rows = list(csv.DictReader(open(path)))
    total = 0.0
    for k in kernels.split(" + "):
        hit = [r for r in rows if r["Name"].startswith(k + "(")]
        if len(hit) != 1:
            return None, None
        total += float(hit[0]["AverageNs"]) * 1e-6
    return total, os.path.relpath(path, ROOT)


def run_leg(name, device, launches, warmup=1, width=None, height=None, want_counts=True):
    """Times `launches` identical launches of one configuration (samples [0, spp) of every pixel) with HIP events on
    the context's stream, and counts box / primitive tests per ray with the instrumented kernels."""
    from raytracer_amd import RenderSettings, _ffi as F
    from raytracer_amd.renderer import Context, Scene, camera_from_settings

    w, h = width or WIDTH, height or HEIGHT
    spp, bg = LEGS[name]
    t0 = time.perf_counter()
    desc, cs, what = leg_scene(name)
    dev = torch.device("cuda", device)
    ctx = Context(device, flags=F.RT_CTX_TIMING)
    ctx.set_max_paths(MAX_PATHS)
    scene = Scene(ctx, desc)
    info = scene.info()
    setup_s = time.perf_counter() - t0
    cam = camera_from_settings(cs, w, h)
    lib = ctx.lib
    rs = RenderSettings(w, h, 0, MAX_DEPTH, spp, bg, F.RT_MODE_GAMMA_PER_SAMPLE, SEED).to_c()
    buf = torch.zeros(w * h * 3, dtype=torch.float32, device=dev)
    torch.cuda.synchronize(dev)
    stats = []
    wall0 = time.perf_counter()
    for i in range(warmup + launches):
        if i == warmup:
            ctx.synchronize()
            wall0 = time.perf_counter()
        st = F.rt_stats()
        F.check(lib, lib.rt_render_device(ctx.handle, scene.handle, C.byref(cam), C.byref(rs), C.c_void_p(buf.data_ptr()),
                                          C.byref(st)))
        if i >= warmup:
            stats.append(st)
    ctx.synchronize()
    wall = time.perf_counter() - wall0
    rays = stats[0].rays_traced
    assert all(s.rays_traced == rays for s in stats), "identical launches must trace identical rays"
    kernel_ms = sum(s.bounce_ms + s.extend_ms + s.shade_ms + s.generate_ms for s in stats) / launches
    total_ms = sum(s.total_ms for s in stats) / launches
    from raytracer_amd.scenes import scene_digest

    out = {"workload": f"{what}; {w}x{h}, depth {MAX_DEPTH}, {spp} spp per launch, progressive (gamma per sample)",
           "scene_sha256": scene_digest(desc),  # tests/golden/scene_digests.json pins the procedural scenes
           "triangles": info.num_triangles, "spheres": info.num_spheres, "bvh_nodes": info.num_nodes, "bvh_depth": info.depth,
           "scene_bytes_in_hbm": info.device_bytes, "scene_setup_s": setup_s,
           "pipeline": PIPELINES.get(stats[0].pipeline, str(stats[0].pipeline)), "launches": launches,
           "rays_per_launch": rays, "paths_per_launch": stats[0].paths, "rays_per_path": rays / stats[0].paths,
           "ms_per_launch": total_ms, "render_kernel_ms_per_launch": kernel_ms, "wall_ms_per_launch": wall / launches * 1e3,
           "mrays_per_s": rays / (total_ms * 1e-3) / 1e6}
    del buf
    scene.close()
    ctx.close()
    if want_counts:
        # box and primitive tests per ray (BVH2-equivalent visits, as the reference's recursion makes them), counted by
        # the instrumented extend kernel on a few samples of the same frame
        cctx = Context(device, flags=F.RT_CTX_COUNTERS)
        cscene = Scene(cctx, desc)
        n_count = 4 if info.num_nodes < 1_000_000 else 1
        crs = RenderSettings(w, h, 0, MAX_DEPTH, n_count, bg, F.RT_MODE_GAMMA_PER_SAMPLE, SEED).to_c()
        cbuf = torch.zeros(w * h * 3, dtype=torch.float32, device=dev)
        st = F.rt_stats()
        F.check(lib, lib.rt_render_device(cctx.handle, cscene.handle, C.byref(cam), C.byref(crs), C.c_void_p(cbuf.data_ptr()),
                                          C.byref(st)))
        n, t, s = (st.node_visits / st.rays_traced, st.triangle_tests / st.rays_traced, st.sphere_tests / st.rays_traced)
        # per ray: 32 B ray + 16 B hit + 32 B per box test + 36 B per triangle test + 16 B per sphere test, plus 80 B of
        # path state per segment
        out.update({"box_tests_per_ray": n, "triangle_tests_per_ray": t, "sphere_tests_per_ray": s,
                    "algorithmic_bytes_per_ray": 32 + 16 + 32 * n + 36 * t + 16 * s + 80})
        del cbuf
        cscene.close()
        cctx.close()
    return out


def roofline_of(name, leg):
    """The roofline of a leg's render kernel.  Scenes whose tables live in LDS or the L2s are bound by VALU issue
    (wave-level instructions per second against one per SIMD every two cycles); the soup is bound by HBM.  The
    instruction and traffic counts come from the committed PMC profile of the same build and the same launch."""
    kernel_hint = {"lanes": "k_paths_lanes", "paths": "k_paths", "paths+lds": "k_paths"}.get(leg["pipeline"], "k_")
    prof, origin = committed_profile(name, kernel_hint, leg["rays_per_launch"])
    secs = leg["render_kernel_ms_per_launch"] * 1e-3
    alg_gbs = leg["algorithmic_bytes_per_ray"] * leg["rays_per_launch"] / secs / 1e9
    hbm_bound = name == "soup10m"
    flops_per_ray = (FLOPS_PER_BOX * leg["box_tests_per_ray"] + FLOPS_PER_TRIANGLE * leg["triangle_tests_per_ray"] +
                     FLOPS_PER_SPHERE * leg["sphere_tests_per_ray"])
    alg_flops = flops_per_ray * leg["rays_per_launch"] / secs
    # the byte formula's figure against the HBM peak: for a scene served by LDS or the L2s these bytes never reach HBM
    alg_key = "algorithmic_frac_of_hbm_peak" if hbm_bound else "algorithmic_bytes_are_lds_or_l2_traffic_ratio_to_hbm_peak"
    r = {"kernel": prof["kernel"] if prof else kernel_hint, "avg_launch_ms": leg["render_kernel_ms_per_launch"],
         "rays_per_launch": leg["rays_per_launch"], "algorithmic_bytes_per_ray": leg["algorithmic_bytes_per_ray"],
         "algorithmic_gbs": alg_gbs, alg_key: alg_gbs / HBM_PEAK_GBS,
         "algorithmic_flops_per_ray": flops_per_ray, "algorithmic_tflops": alg_flops / 1e12,
         "algorithmic_flop_frac": alg_flops / FP32_PEAK_FLOPS,
         "algorithmic_flop_frac_is": f"({FLOPS_PER_BOX} flop x box tests + {FLOPS_PER_TRIANGLE} x triangle tests + "
                                     f"{FLOPS_PER_SPHERE} x sphere tests) per second over {FP32_PEAK_FLOPS / 1e12:.1f} TFLOP/s",
         "pmc": origin if prof else None, "pmc_status": "matches this build and this launch" if prof else origin}
    traffic = None
    if prof:
        d = prof["per_ray"]
        traffic = (d["hbm_read_bytes"] + d["hbm_write_bytes"]) * leg["rays_per_launch"]
        r.update({"hbm_bytes_per_ray": d["hbm_read_bytes"] + d["hbm_write_bytes"],
                  "hbm_traffic_gbs": traffic / secs / 1e9, "hbm_traffic_frac_of_peak": traffic / secs / 1e9 / HBM_PEAK_GBS,
                  "valu_wave_instructions_per_ray": d["valu_wave_instructions"], "lane_utilisation": prof["lane_utilisation"],
                  "l2_requests_per_ray": d.get("l2_read_requests"), "l2_misses_per_ray": d.get("hbm_read_requests"),
                  # the kernel-trace run beside the counter passes: three launches from a cold start (for a launch as
                  # short as Cornell's the clocks are still ramping; the bench command's own summary is below)
                  "avg_launch_ms_in_the_profiled_leg_run": prof.get("avg_dispatch_ms")})
        stats_ms, stats_path = committed_kernel_stats(name, prof["kernel"])
        r.update({"rocprof_kernel_stats_avg_ms": stats_ms, "rocprof_kernel_stats": stats_path})
    if hbm_bound:
        r.update({"bound": "hbm", "achieved": alg_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg_gbs / HBM_PEAK_GBS,
                  "traffic": traffic})
    elif prof:
        ach = prof["per_ray"]["valu_wave_instructions"] * leg["rays_per_launch"] / secs
        r.update({"bound": "valu", "achieved": ach / 1e9, "peak": VALU_PEAK / 1e9, "unit": "G wave-instructions/s",
                  "frac": ach / VALU_PEAK, "traffic": traffic,
                  # an issued instruction counts in `frac` whatever share of its 64 lanes is masked off: this is the
                  # share of the peak's lane-slots that did work
                  "useful_lane_frac": ach / VALU_PEAK * prof["lane_utilisation"],
                  "peak_is": f"{SIMDS} SIMDs x {CLOCK_HZ / 1e9:.1f} GHz / 2 cycles per wave64 VALU instruction "
                             "(= 157.3 TFLOP/s of vector binary32)",
                  "measured_issue_rates": {k: v / 1e9 for k, v in VALU_SUSTAINED.items()},
                  "frac_of_measured_full_rate": ach / VALU_SUSTAINED["full_rate_ops"],
                  # the figure a four-cycle-per-instruction model (SIMD-16 parts; the profiler's VALUBusy) would give
                  "frac_at_four_cycles_per_instruction": ach / (VALU_PEAK / 2.0)})
    else:
        r.update({"bound": "valu", "achieved": None, "peak": VALU_PEAK / 1e9, "unit": "G wave-instructions/s", "frac": None,
                  "useful_lane_frac": None, "traffic": None})
    r["traffic_unit"] = "HBM bytes per launch (PMC: 2 x FETCH_SIZE + WRITE_SIZE; a read request is one 128-byte line)"
    return r


def host_threads():
    """Threads for the CPU baseline: the cores this process may run on (the GPU box gives a job a share of a larger
    machine and says so through OMP_NUM_THREADS)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    share = os.environ.get("OMP_NUM_THREADS", "")
    if share.isdigit() and 0 < int(share) < n:
        n = int(share)
    return max(1, n)


def cpu_baseline(desc, cam_settings):
    """The oracle (CPU restatement of the reference algorithm, NOT the Rust binary) timed on the host cores
    on a bounded sample of the same workload: the same 1920x1080 Cornell frame at reduced spp."""
    from oracle import binding as ob
    from raytracer_amd import RenderSettings, _ffi as F

    threads = host_threads()
    oracle = ob.OracleScene(desc)
    cam = ob.camera_from_settings(cam_settings, WIDTH, HEIGHT)
    rs = RenderSettings(WIDTH, HEIGHT, 0, MAX_DEPTH, 1, mode=F.RT_MODE_GAMMA_PER_SAMPLE, seed=SEED)
    t0 = time.time()
    _, st = oracle.render(cam, rs, threads=threads, rng_mode=ob.RNG_THREAD)
    t1 = time.time() - t0
    spp = int(max(1, min(1024, 15.0 / max(t1, 1e-3))))
    rs = RenderSettings(WIDTH, HEIGHT, 0, MAX_DEPTH, spp, mode=F.RT_MODE_GAMMA_PER_SAMPLE, seed=SEED)
    t0 = time.time()
    _, st = oracle.render(cam, rs, threads=threads, rng_mode=ob.RNG_THREAD)
    dt = time.time() - t0
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = None
    return {"value": st.rays_traced / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "host": {"os_cpu_count": os.cpu_count(), "cpus_in_affinity_mask": affinity,
                     "OMP_NUM_THREADS": os.environ.get("OMP_NUM_THREADS")},
            "sample": f"Cornell {WIDTH}x{HEIGHT}, {spp} spp, depth {MAX_DEPTH}, {st.rays_traced} rays in {dt:.2f} s; "
                      "CPU restatement of the reference algorithm (per-thread xoshiro256++, 256-pixel chunks), "
                      "not the Rust binary"}


def main():
    global WIDTH, HEIGHT
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--spp-per-step", type=int, default=SPP_PER_STEP)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="skip the teapot / spheres / soup legs")
    ap.add_argument("--profile-leg", default=None, help="run three launches of one leg's reference workload and exit")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL); gloo is for rehearsals")
    ap.add_argument("--share-device0", action="store_true",
                    help="rehearsal on a one-GPU box: every rank renders on cuda:0 (use with --backend gloo)")
    ap.add_argument("--total-spp", type=int, default=1024,
                    help="after the timed steps, render exactly this many samples per pixel split over the ranks, "
                         "with one exchange and one download, and report the wall time (0 = skip)")
    ap.add_argument("--dump-image", default=None, help="rank 0 writes the images of the run to this .npz (tests)")
    ap.add_argument("--width", type=int, default=WIDTH)
    ap.add_argument("--height", type=int, default=HEIGHT)
    args = ap.parse_args()

    # `python bench.py --gpus N` without a launcher: start the N ranks ourselves, as a child process and before
    # anything here touches the GPU, and pass its output and exit code on.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        import subprocess

        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE)")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.share_device0 else int(os.environ.get("LOCAL_RANK", "0"))
    WIDTH, HEIGHT = args.width, args.height

    if args.profile_leg:
        torch.cuda.set_device(0)
        leg = run_leg(args.profile_leg, 0, launches=3, warmup=0, want_counts=False)
        leg["source_digest"] = source_digest()
        print(json.dumps({"profile_leg": args.profile_leg, **leg}))
        return

    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
        if dist.get_world_size() != args.gpus:
            sys.exit(f"bench.py: the process group has {dist.get_world_size()} ranks, --gpus asked for {args.gpus}")
    n_gpus = world
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from raytracer_amd import RenderSettings, _ffi as F, load_from_gltf
    from raytracer_amd.distributed import sample_range
    from raytracer_amd.renderer import Context, Scene, camera_from_settings

    desc = load_from_gltf(os.path.join(ROOT, "assets", "cornell.gltf"))
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx = Context(local_rank, stream=stream.cuda_stream, flags=F.RT_CTX_TIMING)
    ctx.set_max_paths(MAX_PATHS)
    scene = Scene(ctx, desc)
    cam = camera_from_settings(desc.camera(0), WIDTH, HEIGHT)
    lib = ctx.lib
    npix = WIDTH * HEIGHT
    spp = args.spp_per_step

    total = torch.zeros(npix * 3, dtype=torch.float32, device=dev)
    # Two step buffers: while step i's accumulator is being all-reduced (RCCL runs on its own stream), step i + 1
    # already renders into the other one.
    step_bufs = [torch.zeros(npix * 3, dtype=torch.float32, device=dev) for _ in range(2)]
    pending = [None, None]
    acc = {"rays": 0, "paths": 0, "render_ms": 0.0, "timed_rays": 0, "timed_launches": 0, "gpu_ms": 0.0,
           "accumulate_ms": 0.0, "pipeline": 0}

    def retire(slot: int):
        """Wait for the all-reduce of buffer `slot` (if any) and fold it into the running total."""
        if pending[slot] is not None:
            pending[slot].wait()
            pending[slot] = None
            total.add_(step_bufs[slot])

    def step(index: int, record: bool):
        # step `index` covers global samples [index * spp * world, (index + 1) * spp * world); rank's block:
        begin, count = sample_range(spp * world, world, rank)
        rs = RenderSettings(WIDTH, HEIGHT, 0, MAX_DEPTH, spp * world, mode=F.RT_MODE_GAMMA_PER_SAMPLE, seed=SEED,
                            sample_begin=index * spp * world + begin, sample_count=count).to_c()
        slot = index & 1
        retire(slot)
        buf = step_bufs[slot]
        buf.zero_()
        st = F.rt_stats()
        F.check(lib, lib.rt_render_device(ctx.handle, scene.handle, C.byref(cam), C.byref(rs),
                                          C.c_void_p(buf.data_ptr()), C.byref(st)))
        if world > 1:
            pending[slot] = dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True)
        else:
            total.add_(buf)
        if record:
            acc["rays"] += st.rays_traced
            acc["paths"] += st.paths
            acc["gpu_ms"] += st.total_ms
            acc["render_ms"] += st.bounce_ms + st.extend_ms + st.shade_ms + st.generate_ms
            acc["accumulate_ms"] += st.accumulate_ms
            acc["timed_rays"] += st.timed_bounce_rays + st.timed_extend_rays
            acc["timed_launches"] += max(1, st.timed_bounce_launches)
            acc["pipeline"] = st.pipeline

    for i in range(args.warmup):
        step(i, False)

    def fence():
        retire(0)
        retire(1)
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i, True)
    fence()
    elapsed = time.perf_counter() - t0

    # Time to N samples per pixel, measured: the ranks split samples [0, N) of every pixel (strong scaling), each renders
    # its block into a zeroed accumulator, ONE all-reduce adds the accumulators and rank 0 downloads the image
    # (src/renderer.rs:105-156 is the loop this stands for).  Wall clock from fence to fence, exchange and download
    # included, maximum over the ranks.
    measured_s, measured_img = None, None
    if args.total_spp > 0:
        host_img = torch.empty(npix * 3, dtype=torch.float32, pin_memory=True)
        begin, count = sample_range(args.total_spp, world, rank)
        rs_total = RenderSettings(WIDTH, HEIGHT, 0, MAX_DEPTH, args.total_spp, mode=F.RT_MODE_GAMMA_PER_SAMPLE, seed=SEED,
                                  sample_begin=begin, sample_count=count).to_c()
        buf = step_bufs[0]
        fence()
        t1 = time.perf_counter()
        buf.zero_()
        if count:
            F.check(lib, lib.rt_render_device(ctx.handle, scene.handle, C.byref(cam), C.byref(rs_total),
                                              C.c_void_p(buf.data_ptr()), None))
        if world > 1:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        if rank == 0:
            host_img.copy_(buf, non_blocking=True)
        fence()
        measured_s = time.perf_counter() - t1
        if rank == 0:
            measured_img = host_img.numpy() / np.float32(args.total_spp)

    if world > 1:
        t = torch.tensor([elapsed, measured_s or 0.0], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0].item())
        measured_s = float(t[1].item()) if measured_s is not None else None
        cnt = torch.tensor([acc["rays"], acc["paths"]], dtype=torch.int64, device=dev)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        rays_all, paths_all = int(cnt[0].item()), int(cnt[1].item())
    else:
        rays_all, paths_all = acc["rays"], acc["paths"]

    if rank == 0:
        img = (total / float(spp * world * (args.steps + args.warmup))).cpu().numpy()
        finite = np.isfinite(img)
        if args.dump_image:
            np.savez(args.dump_image, steps_mean=img.reshape(HEIGHT, WIDTH, 3),
                     total_spp_mean=(measured_img if measured_img is not None else np.zeros(0, np.float32)))
        full_size = (WIDTH, HEIGHT, spp) == (1920, 1080, SPP_PER_STEP)
        scene.close()
        ctx.close()
        del step_bufs, total
        torch.cuda.empty_cache()
        # The headline kernel's roofline.  Timing: HIP events around the render launches of the timed steps, on the
        # context's stream.  Instruction and traffic counts per ray: the committed PMC profile, accepted only if it was
        # taken from these kernel sources on a launch that traced exactly the rays the reference launch (samples
        # [0, spp)) traces in this run.
        ref = run_leg("cornell", local_rank, launches=2, warmup=1, width=WIDTH, height=HEIGHT)
        avg_ms = acc["render_ms"] / max(1, acc["timed_launches"])
        # the timed steps render later sample ranges than the reference launch: their ray counts differ by < 0.1 %
        roof = roofline_of("cornell" if full_size else "rehearsal", {**ref, "render_kernel_ms_per_launch": avg_ms})
        roof.update({"timed_launches": acc["timed_launches"],
                     "rays_per_timed_launch": acc["timed_rays"] / max(1, acc["timed_launches"]),
                     "box_tests_per_ray": ref["box_tests_per_ray"], "triangle_tests_per_ray": ref["triangle_tests_per_ray"],
                     "note": "Cornell's BVH and triangles (4 KB) are staged in LDS: the algorithmic bytes never leave the CU, "
                             "so the kernel is bound by VALU issue, not by HBM; frac = wave-level VALU instructions per "
                             "second over the SIMDs' peak issue rate (which only add/mul/fma-class instructions reach: "
                             "the box test's compares, selects, min and max issue at half of it, see "
                             "measured_issue_rates), algorithmic_gbs is the figure the byte formula gives"})
        out = {
            "metric": "Mrays/s per GPU and per node (1/2/4/8), Cornell 1080p; time to 1024 spp",
            "value": rays_all / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": n_gpus,
            "ranks_in_process_group": dist.get_world_size() if world > 1 else 1,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "assets/cornell.gltf Cornell box, 1920x1080, depth 50, progressive (gamma per "
                                   f"sample), {spp} spp per GPU per step", "width": WIDTH, "height": HEIGHT,
                       "max_depth": MAX_DEPTH, "spp_per_gpu_per_step": spp, "sharding": f"samples x{n_gpus}",
                       "allreduce_floats_per_step": npix * 3 if world > 1 else 0},
            "mrays_per_s_per_gpu": rays_all / elapsed / 1e6 / n_gpus,
            "mpaths_per_s": paths_all / elapsed / 1e6,
            "rays_per_path": rays_all / max(1, paths_all),
            "time_to_1024spp_s": 1024.0 * npix / (paths_all / elapsed),
            "time_to_1024spp_is": "extrapolated from the timed steps' path rate (accumulators stay on the GPUs)",
            f"time_to_{args.total_spp}spp_measured_s": measured_s,
            f"time_to_{args.total_spp}spp_measured_is": (
                f"wall clock of one render of samples [0, {args.total_spp}) of every pixel split over {n_gpus} rank(s) "
                "(strong scaling), one all-reduce of the accumulators and the download of the image to rank 0's host "
                "memory, fence to fence, maximum over ranks") if measured_s is not None else None,
            "pipeline": PIPELINES.get(acc["pipeline"], str(acc["pipeline"])),
            "kernel_ms_rank0": {"gpu_ms": acc["gpu_ms"], "render_kernels_ms": acc["render_ms"],
                                "accumulate_ms": acc["accumulate_ms"]},
            "image_check": {"finite_mean": float(img[finite].mean()), "inf_fraction": float(np.isinf(img).mean()),
                            "nan_fraction": float(np.isnan(img).mean())},
            "roofline": roof,
            "kernel_source_digest": source_digest(),
        }
        if world == 1 and not args.no_legs and full_size:
            legs = {}
            for name in ("teapot", "teapot_lit", "spheres", "soup10m"):
                leg = run_leg(name, local_rank, launches=3 if name != "soup10m" else 2, warmup=1)
                leg["roofline"] = roofline_of(name, leg)
                legs[name] = leg
                torch.cuda.empty_cache()
            out["legs"] = legs
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(desc, desc.camera(0))
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
