"""Frame sharding across ranks (one process per GPU) and reassembly of the
contiguous IQ stream on the rank that feeds the rf_* sink.

Whole frames are independent units. The stream is cut into
blocks of `frames` frames; block b goes to rank b mod world ("block-cyclic"
round-robin). After every rank has rendered its block of a round, the blocks are
gathered to `root` with grouped point-to-point operations: every peer sends its
block straight into its slot of the root's stream buffer, so on an xGMI node
each peer uses its own link to the root; no ring.

Limits of the sharded path (the engine refuses what it cannot shard): SECAM colour and FM video are serial chains
over the whole stream -- those modes render on one rank only. The sound carriers' chain is
serial too, but its state is small: the rank that staged block b - 1 hands it to the rank that stages block b
(sound_state_send / sound_state_recv, a few kilobytes through a host-side group), so every rank runs the chain over
its OWN frames only -- one after the other, as the recurrence demands, but nobody twice.

This module is transport plumbing over torch.distributed; it works with the
`nccl` backend (RCCL on ROCm) on GPUs and with `gloo` on CPU tensors, which is
how tests/test_sharding.py exercises it at world size 2."""
import torch.distributed as dist


def block_of(rank, world, round_index):
    """Index of the block rank `rank` renders in round `round_index`."""
    return round_index * world + rank


def first_frame_of(rank, world, round_index, frames):
    """First frame (0-based) of that block."""
    return block_of(rank, world, round_index) * frames


def gather_start(local, root_buf, rank, world, root=0, group=None):
    """Start reassembling one round of blocks on `root` and return the work handles (wait with gather_wait()):
    the transfers run on the communicator's own stream, so the caller can launch the next round's render
    meanwhile -- as long as it writes another buffer than `local` / `root_buf`."""
    if world == 1:
        return []
    if rank == root:
        if root_buf[root].data_ptr() != local.data_ptr():
            root_buf[root].copy_(local, non_blocking=True)
        ops = [dist.P2POp(dist.irecv, root_buf[r], r, group) for r in range(world) if r != root]
    else:
        ops = [dist.P2POp(dist.isend, local, root, group)]
    return dist.batch_isend_irecv(ops)


def gather_wait(works):
    for w in works:
        w.wait()


def gather_blocks(local, root_buf, rank, world, root=0, group=None, via_host=False):
    """Reassemble one round of blocks on `root`.

    via_host: stage device tensors through host memory (for a `gloo` group on a
    box without RCCL peers; used to dry-run the multi-rank bench on one GPU).

    local     1-D tensor holding this rank's rendered block
    root_buf  on root: tensor [world, local.numel()], row r receives rank r's block
              (root's own row may alias `local`, in which case nothing is copied)
    Returns the list of work handles already waited on (empty for world == 1)."""
    if world == 1:
        return []
    if via_host:
        if rank == root:
            for r in range(world):
                if r == root:
                    if root_buf[root].data_ptr() != local.data_ptr():
                        root_buf[root].copy_(local)
                    continue
                tmp = local.new_empty(local.shape, device="cpu")
                dist.recv(tmp, r, group)
                root_buf[r].copy_(tmp)
        else:
            dist.send(local.cpu(), root, group)
        return []
    if rank == root:
        if root_buf[root].data_ptr() != local.data_ptr():
            root_buf[root].copy_(local)
        ops = [dist.P2POp(dist.irecv, root_buf[r], r, group) for r in range(world) if r != root]
    else:
        ops = [dist.P2POp(dist.isend, local, root, group)]
    works = dist.batch_isend_irecv(ops)
    for w in works:
        w.wait()
    return works


_state_sends = []       # (work, buffer) of the hand-overs on their way: the buffers have to outlive the sends


def sound_state_send(engine, world, block, group=None, last=False):
    """After staging block `block`: hand the sound chains' state to the rank that stages block + 1
    (hvk_sound_state_export). `group`: a host-side (gloo) group -- the state travels as a CPU byte tensor.
    last: nobody stages a block after this one. The send does not wait for its receiver: the next rank may be busy
    receiving THIS rank's rendered block (which this rank has yet to render) before it gets to its next stage."""
    import torch
    if world == 1 or last:
        return
    while _state_sends and _state_sends[0][0].is_completed():
        _state_sends.pop(0)
    buf = torch.frombuffer(bytearray(engine.sound_state_export()), dtype=torch.uint8)
    _state_sends.append((dist.isend(buf, (block + 1) % world, group), buf))


def sound_state_recv(engine, world, block, group=None):
    """Before staging block `block` (> 0): take the chains over from the rank that staged block - 1. Returns the
    position in the 32 kHz source stream the chains go on from (feed the engine's audio queue from there), or None."""
    import torch
    if world == 1 or block == 0:
        return None
    buf = torch.empty(engine.sound_state_size(), dtype=torch.uint8)
    dist.recv(buf, (block - 1) % world, group)
    return engine.sound_state_import(buf.numpy().tobytes())
