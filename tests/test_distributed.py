"""World-size-2 test of the sample sharding on CPU (gloo): two ranks render disjoint sample blocks and one
all-reduce adds their accumulators.  The oracle stands in for the GPU renderer here (test use only)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracer_amd.distributed import render_sharded, sample_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, SPP, SEED = 48, 36, 9, 5


def test_sample_range_partitions_exactly():
    for total in (0, 1, 7, 8, 9, 4096, 1000003):
        for world in (1, 2, 3, 4, 8):
            blocks = [sample_range(total, world, r) for r in range(world)]
            assert blocks[0][0] == 0
            for (b0, c0), (b1, _) in zip(blocks, blocks[1:]):
                assert b0 + c0 == b1
            assert blocks[-1][0] + blocks[-1][1] == total
            counts = [c for _, c in blocks]
            assert max(counts) - min(counts) <= 1
    with pytest.raises(ValueError):
        sample_range(8, 2, 2)


def test_node_split_in_the_library_equals_sample_range(tmp_path):
    """rt_render_node splits a render's samples among its GPUs with rt::sample_block (csrc/sample_split.h).  The native
    check asserts contiguous, disjoint, covering blocks of near-equal length for 1, 2, 3 and 8 GPUs and 0, 1, 7, 8 and 4096
    samples (more GPUs than samples included); its output must be sample_range's blocks, shifted by the first sample."""
    import subprocess

    exe = str(tmp_path / "sample_split_check")
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "raytracer_amd", "csrc"),
                    os.path.join(ROOT, "tests", "native", "sample_split_check.cpp"), "-o", exe], check=True)
    lines = subprocess.run([exe], check=True, capture_output=True, text=True).stdout.strip().splitlines()
    assert len(lines) == 4 * 5 * 3
    for line in lines:
        head, blocks = line.split(":")
        n, total, begin = (int(v) for v in head.split())
        got = [int(v) for v in blocks.split()]
        want = []
        for r in range(n):
            b, c = sample_range(total, n, r)
            want += [begin + b, c]
        assert got == want, line


def _oracle_range_renderer():
    from oracle import binding as ob
    from raytracer_amd import RenderSettings, _ffi as F, load_from_gltf

    desc = load_from_gltf(os.path.join(ROOT, "assets", "cornell.gltf"))
    oracle = ob.OracleScene(desc)
    cam = ob.camera_from_settings(desc.camera(0), W, H)

    def render_range(begin, count, accum):
        rs = RenderSettings(W, H, 0, 50, SPP, mode=F.RT_MODE_GAMMA_PER_SAMPLE, seed=SEED, sample_begin=begin,
                            sample_count=count)
        buf = accum.numpy().reshape(H, W, 3)
        oracle.render(cam, rs, threads=2, accum=buf)

    return render_range


def _worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    accum = torch.zeros(H * W * 3, dtype=torch.float32)
    render_sharded(_oracle_range_renderer(), SPP, accum, world, rank)
    if rank == 0:
        np.save(out_path, accum.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_one(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "sum.npy")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    two = np.load(out)
    one = torch.zeros(H * W * 3, dtype=torch.float32)
    render_sharded(_oracle_range_renderer(), SPP, one, 1, 0)
    one = one.numpy()
    # same samples, different association of the float additions: (s0+..+s4) + (s5+..+s8) vs s0+..+s8
    assert np.array_equal(np.isinf(one), np.isinf(two)) and np.array_equal(np.isnan(one), np.isnan(two))
    fin = np.isfinite(one)
    assert np.allclose(one[fin], two[fin], rtol=2e-6, atol=1e-6)
    assert one[fin].sum() > 0
