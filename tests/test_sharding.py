"""The N > 1 path on CPU: two gloo ranks shard blocks of frames, each 'renders'
its block with the oracle standing in for the GPU engine (test infrastructure),
and the gathered stream on rank 0 must equal the single-process stream."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _render_block(first_frame, frames):
    """Frames [first_frame, first_frame + frames) of the PAL-I stream via the oracle
    (serial: it has to walk the stream from frame 0, like the host audio pre-pass)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    import util
    g = util.Golden()
    conf, sr = g.conf("i_full")
    with oracle.Oracle(conf, sr) as o:
        o.set_frame(g.frame("i_full"))
        o.set_audio(g.audio, True)
        if first_frame:
            o.render_lines(625 * first_frame)
        return o.render_lines(625 * frames)


def _worker(rank, world, port, frames, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from hacktv_amd import sharding

    first = sharding.first_frame_of(rank, world, 0, frames)
    local = torch.from_numpy(_render_block(first, frames).reshape(-1).copy())
    root_buf = torch.empty((world, local.numel()), dtype=torch.int16) if rank == 0 else None
    sharding.gather_blocks(local, root_buf, rank, world)
    dist.barrier()
    if rank == 0:
        np.save(out_path, root_buf.numpy())
    dist.destroy_process_group()


def test_two_ranks_reassemble_the_contiguous_stream(tmp_path):
    world, frames = 2, 1
    out = str(tmp_path / "stream.npy")
    mp.spawn(_worker, args=(world, _free_port(), frames, out), nprocs=world, join=True)
    got = np.load(out).reshape(-1, 2)
    want = _render_block(0, world * frames)
    assert np.array_equal(got, want)


def test_block_assignment_is_round_robin():
    sys.path.insert(0, ROOT)
    from hacktv_amd import sharding
    world, frames = 8, 128
    seen = []
    for rnd in range(3):
        for r in range(world):
            seen.append(sharding.first_frame_of(r, world, rnd, frames))
    assert seen == [i * frames for i in range(3 * world)]
