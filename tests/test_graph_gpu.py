"""pgp_score_lcp_device is allocation-free after pgp_reserve and enqueues on the caller's stream only,
so a caller can capture it in a HIP graph: capture one scoring call with
torch.cuda.CUDAGraph, replay it on new transforms written into the captured input buffer, and
compare with ordinary calls."""
import numpy as np
import pytest

from physimglobalpose_amd import LcpScorer, PGP_MODE_PLAIN, PGP_MODE_WEIGHTED, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", [PGP_MODE_PLAIN, PGP_MODE_WEIGHTED])
def test_scoring_call_replays_from_a_graph(mode):
    import torch
    w = synth.make_workload(6000, 900, 3 * 512, config_id=61)
    sc = LcpScorer(0)
    sc.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
    sc.reserve(512)
    batches = [torch.from_numpy(w.T[k * 512:(k + 1) * 512]).cuda() for k in range(3)]
    d_T = batches[0].clone()
    d_s = torch.zeros(512, device="cuda")
    d_c = torch.zeros(512, dtype=torch.int32, device="cuda")
    d_b = torch.zeros(2, dtype=torch.int32, device="cuda")
    expect = []
    for b in batches:                                   # ordinary calls first
        sc.score_device(b, d_s, d_c, d_b, mode=mode, gate_deg=w.gate_deg)
        torch.cuda.synchronize()
        expect.append((d_s.cpu().numpy().copy(), d_c.cpu().numpy().copy(), d_b.cpu().numpy().copy()))
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        sc.score_device(d_T, d_s, d_c, d_b, mode=mode, gate_deg=w.gate_deg)      # warm-up on the capture stream
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=side):
            sc.score_device(d_T, d_s, d_c, d_b, mode=mode, gate_deg=w.gate_deg, stream=side)
    torch.cuda.current_stream().wait_stream(side)
    for k in (1, 2, 0, 2):
        d_T.copy_(batches[k])
        g.replay()
        torch.cuda.synchronize()
        s, c, b = expect[k]
        assert np.array_equal(d_s.cpu().numpy(), s)
        assert np.array_equal(d_c.cpu().numpy(), c)
        assert np.array_equal(d_b.cpu().numpy(), b)


def test_capture_after_one_host_pointer_warm_up_on_a_small_scene():
    """A small scene builds its index on the context's side stream: set_scene -> ONE synchronous host-pointer
    scoring call -> graph capture must work (the synchronous call's wait clears the pending flag)."""
    import torch
    w = synth.make_workload(2500, 700, 256, config_id=62)
    sc = LcpScorer(0)
    sc.set_model(w.Q_xyz, w.Q_nrm)
    sc.reserve(256)
    for rep in range(3):
        sc.set_scene(w.P_xyz, w.P_nrm, w.P_w, w.delta)           # side-stream build queued, call returns
        s, c, bi, bs = sc.score(w.T, PGP_MODE_WEIGHTED, w.gate_deg)   # waits on the device, synchronises the host
        d_T = torch.from_numpy(w.T).cuda()
        d_s = torch.zeros(256, device="cuda")
        d_c = torch.zeros(256, dtype=torch.int32, device="cuda")
        d_b = torch.zeros(2, dtype=torch.int32, device="cuda")
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                sc.score_device(d_T, d_s, d_c, d_b, mode=PGP_MODE_WEIGHTED, gate_deg=w.gate_deg, stream=side)
        torch.cuda.current_stream().wait_stream(side)
        g.replay()
        torch.cuda.synchronize()
        assert np.array_equal(d_s.cpu().numpy(), s) and np.array_equal(d_c.cpu().numpy(), c)
        assert int(d_b[0]) == bi
