#!/usr/bin/env python3
"""Benchmark of the hot path: Mrays/s on the Cornell box at 1920x1080, and the time to 1024 samples per pixel.

One "step" renders SPP_PER_STEP samples of every pixel on each GPU (generate / extend / shade / accumulate,
all bounces) and — when more than one GPU takes part — adds the step's radiance accumulators with one RCCL
all-reduce.  Inputs (scene tables, camera) are resident in HBM before the timed region; the accumulators stay
on the device.  One ray = one closest-hit query (`world.hit`, src/renderer.rs:44 of the reference).

    python bench.py [--gpus N] [--steps K] [--warmup W]
N > 1 runs one rank per GPU: either a launcher (torch.distributed.run) has started the N ranks, or bench.py
starts them itself as a child process.
"""
from __future__ import annotations

import argparse
import ctypes as C
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
MAX_PATHS = 1 << 27          # path slots per batch: the 64 samples of a 1080p frame a step renders (11.8 GB of state)
TIMED_BOUNCES = 6            # extend launches bracketed by HIP events per batch (they carry > 98 % of the rays)
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E peak


def cpu_baseline(desc, cam_settings, threads):
    """The oracle (CPU restatement of the reference algorithm, NOT the Rust binary) timed on the host cores
    on a bounded sample of the same workload: the same 1920x1080 Cornell frame at reduced spp."""
    from oracle import binding as ob
    from raytracer_amd import RenderSettings, _ffi as F

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
    return {"value": st.rays_traced / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
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
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL); gloo is for rehearsals")
    ap.add_argument("--share-device0", action="store_true",
                    help="rehearsal on a one-GPU box: every rank renders on cuda:0 (use with --backend gloo)")
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

    WIDTH, HEIGHT = args.width, args.height

    from raytracer_amd import RenderSettings, _ffi as F, load_from_gltf
    from raytracer_amd.distributed import sample_range
    from raytracer_amd.renderer import Context, Scene, camera_from_settings

    desc = load_from_gltf(os.path.join(ROOT, "assets", "cornell.gltf"))
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx = Context(local_rank, stream=stream.cuda_stream, flags=F.RT_CTX_TIMING)
    ctx.set_max_paths(MAX_PATHS)
    ctx.set_timed_bounces(TIMED_BOUNCES)
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
    acc = {"rays": 0, "paths": 0, "bounce_ms": 0.0, "extend_ms": 0.0, "timed_rays": 0, "timed_launches": 0,
           "gpu_ms": 0.0, "shade_ms": 0.0, "generate_ms": 0.0, "accumulate_ms": 0.0, "pipeline": 0,
           "first_ms": 0.0, "first_rays": 0, "first_launches": 0}

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
            acc["bounce_ms"] += st.bounce_ms
            acc["extend_ms"] += st.extend_ms
            acc["shade_ms"] += st.shade_ms
            acc["generate_ms"] += st.generate_ms
            acc["accumulate_ms"] += st.accumulate_ms
            acc["timed_rays"] += st.timed_bounce_rays
            acc["timed_launches"] += st.timed_bounce_launches
            acc["pipeline"] = st.pipeline
            acc["first_ms"] += st.first_bounce_ms
            acc["first_rays"] += st.first_bounce_rays
            acc["first_launches"] += st.first_bounce_launches

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

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        cnt = torch.tensor([acc["rays"], acc["paths"]], dtype=torch.int64, device=dev)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        rays_all, paths_all = int(cnt[0].item()), int(cnt[1].item())
    else:
        rays_all, paths_all = acc["rays"], acc["paths"]

    if rank == 0:
        # per-ray algorithmic bytes of the extend kernel from an instrumented pass (outside the timed region):
        # 32 B ray read + 16 B hit write + 32 B per box test + 36 B per triangle test + 16 B per sphere test.
        cctx = Context(local_rank, flags=F.RT_CTX_COUNTERS)
        cscene = Scene(cctx, desc)
        rs = RenderSettings(WIDTH, HEIGHT, 0, MAX_DEPTH, 4, mode=F.RT_MODE_GAMMA_PER_SAMPLE, seed=SEED).to_c()
        buf = torch.zeros(npix * 3, dtype=torch.float32, device=dev)
        st = F.rt_stats()
        F.check(lib, lib.rt_render_device(cctx.handle, cscene.handle, C.byref(cam), C.byref(rs),
                                          C.c_void_p(buf.data_ptr()), C.byref(st)))
        nodes_per_ray = st.node_visits / st.rays_traced
        tris_per_ray = st.triangle_tests / st.rays_traced
        sph_per_ray = st.sphere_tests / st.rays_traced
        # + 80 B of path state per segment (throughput, radiance, pixel, depth/flags, rng; read and written)
        bytes_per_ray = 32 + 16 + 32 * nodes_per_ray + 36 * tris_per_ray + 16 * sph_per_ray + 80
        cscene.close()
        cctx.close()

        # HBM bytes per ray of the same kernel from the committed PMC run (FETCH_SIZE x2 + WRITE_SIZE)
        traffic_per_ray = None
        tj = os.path.join(ROOT, "profiles", "r01_pmc_traffic_cornell1080p.json")
        if os.path.exists(tj):
            kernels = json.load(open(tj)).get("kernels", {})
            kern = kernels.get("k_paths") or kernels.get("k_bounce")
            if kern:
                traffic_per_ray = kern["hbm_read_bytes_per_ray"] + kern["hbm_write_bytes_per_ray"]

        # VALU picture of the same kernel from the committed PMC run (the kernel is instruction-issue bound, which the
        # HBM roofline above cannot show): wave-level VALU instructions per ray, VALU-busy share of SIMD cycles
        # (4 cycles per wave64 instruction, 1024 SIMDs, GRBM_GUI_ACTIVE summed over the 8 XCDs) and lane utilisation.
        valu = None
        pj = os.path.join(ROOT, "profiles", "r01_v8_pmc_summary_cornell1080p_paths_lds.txt")
        if os.path.exists(pj) and os.path.exists(tj):
            vals, on = {}, False
            for line in open(pj):
                if not line.startswith(" "):
                    on = "k_paths" in line
                elif on:
                    parts = line.split()
                    vals[parts[0]] = float(parts[1])
            pmc_rays = json.load(open(tj)).get("rays", 0)
            if pmc_rays and "SQ_INSTS_VALU" in vals:
                simd_cycles = vals["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
                valu = {"wave_instructions_per_ray": vals["SQ_INSTS_VALU"] / pmc_rays,
                        "busy_share_of_simd_cycles": 4.0 * vals["SQ_ACTIVE_INST_VALU"] / simd_cycles,
                        "lane_utilisation": vals["SQ_THREAD_CYCLES_VALU"] / (64.0 * vals["SQ_INSTS_VALU"]),
                        "source": "profiles/r01_v8_pmc_summary_cornell1080p_paths_lds.txt (PMC, not measured in this run; "
                                  "the busy share divides counters of two profiling passes and reads slightly above 1 when "
                                  "the VALUs are saturated)"}

        # The fused kernel has two instances: k_bounce<FIRST=true> (bounce 0: camera rays generated in place) and
        # k_bounce<FIRST=false> (later bounces).  Later bounces dominate the time; bounces 1..TIMED_BOUNCES-1 of every
        # batch are bracketed by events here, bounce 0 separately.
        def group(ms, rays, launches):
            avg_ms = ms / max(1, launches)
            per_launch = rays / max(1, launches)
            ach = bytes_per_ray * per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
            return avg_ms, per_launch, ach

        later = group(acc["bounce_ms"] - acc["first_ms"], acc["timed_rays"] - acc["first_rays"],
                      acc["timed_launches"] - acc["first_launches"])
        first = group(acc["first_ms"], acc["first_rays"], acc["first_launches"])
        avg_launch_ms, rays_per_launch, achieved = later
        one_launch_per_batch = acc["pipeline"] in (F.RT_PIPELINE_PATHS, F.RT_PIPELINE_PATHS_LDS)
        if one_launch_per_batch:
            kernel_name = ("k_paths<true,false> (one launch per batch of %d spp: generate + extend + shade of every bounce; "
                           "BVH and triangles staged in LDS)" % min(spp, MAX_PATHS // npix))
        else:
            kernel_name = ("k_bounce<false,true,false> (extend + shade of bounces 1..%d of each batch, BVH and triangles "
                           "staged in LDS)" % (TIMED_BOUNCES - 1))
        img = (total / float(spp * world * (args.steps + args.warmup))).cpu().numpy()
        finite = np.isfinite(img)
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
            "pipeline": {2: "fused", 3: "fused+lds", 7: "paths", 8: "paths+lds"}.get(acc["pipeline"], str(acc["pipeline"])),
            "kernel_ms_rank0": {k: acc[k] for k in ("gpu_ms", "bounce_ms", "generate_ms", "extend_ms", "shade_ms",
                                                    "accumulate_ms")},
            "image_check": {"finite_mean": float(img[finite].mean()), "inf_fraction": float(np.isinf(img).mean()),
                            "nan_fraction": float(np.isnan(img).mean())},
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": None if traffic_per_ray is None else traffic_per_ray * rays_per_launch,
                "traffic_unit": "bytes per launch (HBM, PMC: 2 x FETCH_SIZE + WRITE_SIZE, profiles/r01_pmc_traffic_cornell1080p.json)",
                "kernel": kernel_name,
                "avg_launch_ms": avg_launch_ms, "rays_per_launch": rays_per_launch,
                "first_bounce_kernel": None if one_launch_per_batch else {
                    "kernel": "k_bounce<true,true,false> (generate + extend + shade of bounce 0)",
                    "avg_launch_ms": first[0], "rays_per_launch": first[1], "achieved": first[2],
                    "frac": first[2] / HBM_PEAK_GBS, "launches": acc["first_launches"]},
                "valu": valu,
                "algorithmic_bytes_per_ray": bytes_per_ray, "nodes_per_ray": nodes_per_ray,
                "triangle_tests_per_ray": tris_per_ray, "timed_launches": acc["timed_launches"] - acc["first_launches"],
                "note": "Cornell's BVH + triangles are 4 KB and are read from LDS: of the algorithmic bytes only the path "
                        "state reaches HBM (see traffic), so frac is an on-chip bandwidth figure quoted against the HBM "
                        "peak as the contract asks; the kernel is VALU-issue bound (profiles/r01_pmc_summary_*.txt)",
            },
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(desc, desc.camera(0), threads=min(16, os.cpu_count() or 1))
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
