"""The N > 1 path on CPU: two gloo ranks shard blocks of frames, each 'renders'
its block with the oracle standing in for the GPU engine (test infrastructure),
and the gathered stream on rank 0 must equal the single-process stream."""
import os
import socket
import sys

import numpy as np
import pytest
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


# ---- the sound chains handed from rank to rank (hvk_sound_state_export / _import) ----

def _sound_worker(rank, world, port, frames, rounds, modes, out_dir):
    """Each rank runs the host half of the engine (device = -1: tables + the serial sound chains, no GPU) over ITS blocks
    only: before a block it takes the chains' state from the rank that did the block before, after it it hands it on."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import hacktv_amd as H
    from hacktv_amd import sharding
    import util
    g = util.Golden()
    for mode, flags, members in modes:
        conf = H.preset(mode, flags)
        for k, v in members.items():
            setattr(conf, k, v)
        with H.Engine(conf, 16000000 if mode != "m" else 13500000, device=-1) as e:
            fs, prime = e.info["frame_samples"], e.info["startup_samples"]
            L = len(g.audio)
            mine = {}
            for rnd in range(rounds):
                b = sharding.block_of(rank, world, rnd)
                pos = sharding.sound_state_recv(e, world, b)
                if pos is not None:
                    # the queue is empty now and the next write is taken to start at source position `pos`:
                    # the test tone is a loop of L samples
                    e.audio_write(g.audio[pos % L:])
                for _ in range(2 + frames):
                    e.audio_write(g.audio)
                car, sym, k0 = e.host_side_streams(b * frames * fs + prime, frames * fs)
                sharding.sound_state_send(e, world, b, last=(rnd == rounds - 1 and rank == world - 1))
                mine[b] = (car, sym, k0)
            np.savez(os.path.join(out_dir, "%s_%d.npz" % (mode, rank)), generated=e.sound_samples_generated(),
                     **{"car%d" % b: v[0] for b, v in mine.items()}, **{"sym%d" % b: v[1] for b, v in mine.items()},
                     **{"k0_%d" % b: v[2] for b, v in mine.items()})
    dist.barrier()
    dist.destroy_process_group()


def test_sound_chains_go_from_rank_to_rank(tmp_path):
    """2 ranks x 3 rounds: the serial sound chains (FM carrier with limiter + NICAM for PAL-I; the AM carrier + NICAM of
    system L; both FM carriers, pilot and identification tone of A2 stereo) run over every block exactly once, on the
    rank that renders it, from the state the block before left -- and the side streams every rank hands to its GPU
    (carrier samples, NICAM symbols) equal those of ONE engine that ran over the whole stream. The counter says each
    rank's chains worked through its own frames only (plus, on rank 0, the pipeline's start-up samples)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import hacktv_amd as H
    import util
    g = util.Golden()
    world, frames, rounds = 2, 2, 3
    modes = [("i", H.FLAG_FILTER, {}), ("l", H.FLAG_FILTER, {}), ("g", H.FLAG_FILTER, {"a2stereo": 1})]
    mp.spawn(_sound_worker, args=(world, _free_port(), frames, rounds, modes, str(tmp_path)), nprocs=world, join=True)
    for mode, flags, members in modes:
        conf = H.preset(mode, flags)
        for k, v in members.items():
            setattr(conf, k, v)
        with H.Engine(conf, 16000000, device=-1) as e:
            fs, prime = e.info["frame_samples"], e.info["startup_samples"]
            for _ in range(2 + world * rounds * frames):
                e.audio_write(g.audio)
            got = [np.load(os.path.join(str(tmp_path), "%s_%d.npz" % (mode, r))) for r in range(world)]
            for b in range(world * rounds):
                car, sym, k0 = e.host_side_streams(b * frames * fs + prime, frames * fs)
                r = got[b % world]
                assert np.array_equal(r["car%d" % b], car), "%s: carriers of block %d" % (mode, b)
                assert int(r["k0_%d" % b]) == k0 and np.array_equal(r["sym%d" % b], sym), "%s: NICAM symbols of block %d" % (mode, b)
            for rk in range(world):
                own = rounds * frames * fs
                assert int(got[rk]["generated"]) == own + (prime if rk == 0 else 0), "%s: rank %d ran its chains over %d samples" % (mode, rk, int(got[rk]["generated"]))


def test_sound_state_is_refused_by_another_configuration():
    sys.path.insert(0, ROOT)
    import hacktv_amd as H
    with H.Engine(H.preset("i", H.FLAG_FILTER), 16000000, device=-1) as a, H.Engine(H.preset("l", H.FLAG_FILTER), 16000000, device=-1) as b, \
            H.Engine(H.preset("i", H.FLAG_FILTER | H.FLAG_NOAUDIO), 16000000, device=-1) as c:
        st = a.sound_state_export()
        assert len(st) == a.sound_state_size() > 0
        with pytest.raises(H.HvkError):
            b.sound_state_import(st)                 # AM + NICAM is not FM + NICAM
        with pytest.raises(H.HvkError):
            b.sound_state_import(st[:100])
        assert c.sound_state_size() == 0
        with pytest.raises(H.HvkError):
            c.sound_state_import(st)                 # --noaudio: no chains
