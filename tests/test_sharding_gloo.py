"""CPU, world_size 2 over gloo: the frame-sharded clip driver assembles frames in order on rank 0
(the same host code runs over RCCL on the GPU box; only the backend differs)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _fake_frame(t, hw):
    # deterministic function of the GLOBAL frame index only
    f = torch.full((hw[0], hw[1], 3), t % 251, dtype=torch.uint8)
    f[0, 0, 0] = (t * 7) % 256
    return f


def _worker(rank, world, port, T, hw, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from real3dportrait_amd.frames import render_clip_sharded, shard_frames
        rendered = []

        def render(t):
            rendered.append(t)
            return _fake_frame(t, hw)

        clip = render_clip_sharded(render, T, frame_hw=hw, device="cpu")
        lo, hi = shard_frames(T, world, rank)
        assert rendered == list(range(lo, hi))            # each rank renders exactly its contiguous chunk
        if rank == 0:
            assert clip.shape == (T, hw[0], hw[1], 3) and clip.dtype == torch.uint8
            for t in range(T):
                assert torch.equal(clip[t], _fake_frame(t, hw)), t
        else:
            assert clip is None
        q.put((rank, "ok"))
    except Exception as e:          # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("T", [10, 7, 1])
def test_sharded_clip_gloo_world2(T):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, T, (8, 6), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_single_process_path():
    from real3dportrait_amd.frames import render_clip_sharded
    clip = render_clip_sharded(lambda t: _fake_frame(t, (4, 4)), 5, frame_hw=(4, 4), device="cpu")
    assert clip.shape == (5, 4, 4, 3) and int(clip[3, 1, 1, 1]) == 3
