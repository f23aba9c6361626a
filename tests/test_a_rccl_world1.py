"""The device-buffer branch of the strips' exchange on RCCL, at world 1 on the one GPU of the box.

Every multi-rank test of ``StripExchange`` runs on gloo with host staging (tests/test_dist_cpu.py)
and a one-rank run issues no collective at all, so without this file not one line of the branch
that an 8-GPU run takes - ``dist.gather(packed, views of one buffer, async_op=True)`` /
``dist.reduce(uint8, SUM, async_op=True)`` issued from a lane's stream on the lane's own
communicator, ``work.wait()``, the copies behind it - would ever have executed before the first
hardware run.  It proves nothing about xGMI; it executes the code.

(The file sorts first so that the process group comes up before the session's other tests have
touched the GPU; nothing is re-exec'ed.)"""
import socket

import pytest


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_strip_exchange_on_device_buffers_through_rccl_at_world_1():
    import torch
    import torch.distributed as dist
    from pano360_amd import dist as pdist
    from pano360_amd import engine, synth

    assert torch.cuda.is_available()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0,
                            world_size=1, device_id=torch.device("cuda:0"))
    try:
        assert dist.get_backend() == "nccl"
        imgs, rots, intrs = synth.make_scene(6, 320, 180, sweep_deg=110.0, jitter=0.01, seed=11,
                                             kind="B")
        shapes = [im.shape[:2] for im in imgs]
        lanes = [engine.Engine("cuda:0"), engine.Engine("cuda:0")]
        frames = lanes[0].upload_frames(imgs)
        plan = lanes[0].upload_plan(engine.Plan(shapes, rots, intrs, True, 10 ** 9))
        want = lanes[0].stitch(frames, plan, "multiband", 5)[0].clone()
        torch.cuda.synchronize()
        for mode, lane_groups in [(m, g) for m in pdist.StripExchange.MODES for g in ("shared", "own")]:
            st = pdist.ShardedStitcher(lanes, shapes, rots, intrs, 5, 0, 1, exchange=mode, depth=2,
                                       force_collective=True, lane_groups=lane_groups)
            for _, _, ex in st.lanes:
                assert ex.collective and not ex.host_staged
            if lane_groups == "own":                                     # a communicator per lane
                assert st.lanes[0][2].group is not None
                assert st.lanes[0][2].group is not st.lanes[1][2].group
            else:                                                        # one, the default group's
                assert st.lanes[0][2].group is None and st.lanes[1][2].group is None
            got = []
            for _ in range(6):                                           # two lanes, depth 2
                previous = st.step(frames)[1]
                if previous is not None:
                    got.append(previous.clone())
            got.append(st.finish().clone())
            torch.cuda.synchronize()
            assert len(got) == 6
            for k, mosaic in enumerate(got):
                assert torch.equal(mosaic, want), f"{mode} / {lane_groups}: stitch {k} differs from Engine.stitch"
            st.close()
        # the timing reduction and the job description of bench.py on the same backend
        assert pdist.max_over_ranks(1.5, "cuda:0") == 1.5
        seen = pdist.describe_job("cuda:0", "cuda:0")
        assert seen["world_size"] == 1 and seen["backend"] == "nccl"
        assert seen["allreduce_checksum"] == seen["allreduce_expected"] == 1
    finally:
        dist.destroy_process_group()
