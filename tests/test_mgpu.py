"""sn_mgpu_*: the multi-GPU form of the C ABI.  CPU: the shard arithmetic (pure) agrees with the torch.distributed
form (dist.shard_range) for every (n, ndev) and the entry points fail loudly without a GPU.  GPU: with ndev = 1 both
forms (host buffers; device-resident shards gathered on device 0) equal sn_infer_batch bit for bit; more devices than
the box has is an argument error, never a silent smaller job."""
import numpy as np
import pytest

from hobot_stereonet_amd import api, dist as sdist, synth


def test_shard_arithmetic_matches_dist():
    for ndev in (1, 2, 3, 4, 7, 8):
        for n in (0, 1, 5, 8, 63, 64, 65, 512):
            covered = 0
            for k in range(ndev):
                first, count = api.mgpu_shard(n, ndev, k)
                b, e = sdist.shard_range(n, k, ndev)
                assert (first, first + count) == (b, e)
                assert first == covered
                covered += count
            assert covered == n
    with pytest.raises(api.StereoNetError):
        api.mgpu_shard(4, 2, 2)
    with pytest.raises(api.StereoNetError):
        api.mgpu_shard(4, 0, 0)


def test_ticket_ring_state_machine():
    """The double buffering of sn_mgpu_submit_device / sn_mgpu_wait: tickets count from 1, ticket t owns slot t % 2 from
    submit to wait, two may be in flight, waits may come in either order, a submit whose slot is still owned is BUSY, a
    stale or unknown ticket is refused."""
    r = api.MgpuRing()
    t1, s1 = r.submit()
    t2, s2 = r.submit()
    assert (t1, t2) == (1, 2) and {s1, s2} == {0, 1}
    with pytest.raises(api.StereoNetError) as e:
        r.submit()
    assert e.value.code == -6                      # SN_ERR_BUSY: both slots in flight
    assert r.wait(t2) == s2                        # out of order
    with pytest.raises(api.StereoNetError) as e:
        r.wait(t2)
    assert e.value.code == -7                      # SN_ERR_TICKET: consumed
    with pytest.raises(api.StereoNetError) as e:   # ticket 3 wants ticket 1's slot: waiting for ticket 2 did not free it
        r.submit()
    assert e.value.code == -6
    assert r.wait(t1) == s1
    t3, s3 = r.submit()
    t4, s4 = r.submit()
    assert (t3, s3, t4, s4) == (3, s1, 4, s2)
    with pytest.raises(api.StereoNetError):
        r.wait(99)
    r.wait(t3)
    for k in range(5, 12):                         # steady state: submit k, then wait for k - 1
        t, s = r.submit()
        assert t == k and s == k % 2
        r.wait(k - 1)


def test_create_without_gpu_fails_loudly(tmp_path, weights_blob):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from hobot_stereonet_amd import weights
    p = str(tmp_path / "m.snw")
    weights.save_snw(p, weights_blob)
    with pytest.raises(api.StereoNetError) as e:
        api.StereoNetMultiGPU(p, ndev=1)
    assert e.value.code == -4        # SN_ERR_DEVICE


@pytest.mark.gpu
def test_one_device_equals_single_gpu_batch(model_factory):
    import torch
    w, h, d = 160, 96, 96
    n = 5
    xs = np.stack([synth.model_input_i8(w, h, d, 70 + s) for s in range(n)])
    with api.StereoNetHIP(model_factory(w, h, d), max_batch=n, precision=api.PREC_F16) as eng:
        disp1, raw1 = eng.infer(xs)
    with api.StereoNetMultiGPU(model_factory(w, h, d), devices=[0], max_batch=n, precision=api.PREC_F16) as m:
        assert m.ndev == 1 and m.per_device_batch == n and m.gather_kind == 1
        disp, raw = m.infer(xs)
        assert (raw == raw1).all() and (disp == disp1).all()
        disp_b, raw_b = m.infer(xs[:3])                  # a smaller batch through the same object
        assert (raw_b == raw1[:3]).all() and (disp_b == disp1[:3]).all()
        # device form: shard 0 resident on device 0, maps gathered (here: written in place) on device 0
        dev = torch.device("cuda", 0)
        x = torch.from_numpy(xs).to(dev)
        traw = torch.empty((n, h, w), dtype=torch.int32, device=dev)
        tdisp = torch.empty((n, h, w), dtype=torch.float32, device=dev)
        m.infer_device(n, [x.data_ptr()], traw.data_ptr(), tdisp.data_ptr())
        torch.cuda.synchronize()
        assert (traw.cpu().numpy() == raw1).all() and (tdisp.cpu().numpy() == disp1).all()
        with pytest.raises(api.StereoNetError):
            m.infer(np.concatenate([xs, xs]))             # n > max_batch


@pytest.mark.gpu
def test_rccl_branch_loads_and_runs_with_one_device(model_factory, monkeypatch):
    """SN_MGPU_GATHER=rccl with ndev = 1: librccl is dlopen'ed, ncclCommInitAll builds the (one-rank) communicator and every
    batch runs its grouped exchange (empty: the root has no peer) on the exchange stream — the RCCL plumbing of sn_mgpu_*
    executes on a one-GPU box; results equal sn_infer_batch."""
    import torch
    monkeypatch.setenv("SN_MGPU_GATHER", "rccl")
    w, h, d, n = 160, 96, 96, 3
    xs = np.stack([synth.model_input_i8(w, h, d, 110 + s) for s in range(n)])
    with api.StereoNetHIP(model_factory(w, h, d), max_batch=n, precision=api.PREC_F16) as eng:
        disp1, raw1 = eng.infer(xs)
    with api.StereoNetMultiGPU(model_factory(w, h, d), devices=[0], max_batch=n, precision=api.PREC_F16) as m:
        assert m.ndev == 1 and m.gather_kind == 2
        dev = torch.device("cuda", 0)
        x = torch.from_numpy(xs).to(dev)
        traw = torch.empty((n, h, w), dtype=torch.int32, device=dev)
        tdisp = torch.empty((n, h, w), dtype=torch.float32, device=dev)
        for _ in range(2):
            m.infer_device(n, [x.data_ptr()], traw.data_ptr(), tdisp.data_ptr())
        assert (traw.cpu().numpy() == raw1).all() and (tdisp.cpu().numpy() == disp1).all()


@pytest.mark.gpu
def test_two_shards_on_one_device_through_the_worker_threads(model_factory, monkeypatch):
    """ndev = 2 with the real engine on a one-GPU box: SN_MGPU_ALLOW_DUP=1 (test switch) lets both shards name device 0, so
    the per-shard worker threads, the shard arithmetic (ragged: 3 + 2), the status agreement, the peer-copy gather into
    the root buffers and the double-buffered submit / wait all run with ndev > 1.  Bit-equal to sn_infer_batch."""
    import torch
    monkeypatch.setenv("SN_MGPU_ALLOW_DUP", "1")
    w, h, d = 160, 96, 96
    n = 5
    xs = np.stack([synth.model_input_i8(w, h, d, 80 + s) for s in range(n)])
    ys = np.stack([synth.model_input_i8(w, h, d, 90 + s) for s in range(n)])
    with api.StereoNetHIP(model_factory(w, h, d), max_batch=n, precision=api.PREC_F16) as eng:
        disp_x, raw_x = eng.infer(xs)
        disp_y, raw_y = eng.infer(ys)
    with api.StereoNetMultiGPU(model_factory(w, h, d), devices=[0, 0], max_batch=n, precision=api.PREC_F16) as m:
        assert m.ndev == 2 and m.per_device_batch == 3 and m.gather_kind == 1      # duplicates: peer copies, not RCCL
        # more than one shard: the library itself puts the engines' pipelines on high-priority streams (its gather runs
        # beside them); the caller does not have to know about SN_STREAM_PRIORITY
        assert m.engine_stream_priority_high(0) and m.engine_stream_priority_high(1)
        disp, raw = m.infer(xs)                                                    # host form
        assert (raw == raw_x).all() and (disp == disp_x).all()
        dev = torch.device("cuda", 0)
        tx, ty = torch.from_numpy(xs).to(dev), torch.from_numpy(ys).to(dev)
        out = [(torch.empty((n, h, w), dtype=torch.int32, device=dev), torch.empty((n, h, w), dtype=torch.float32, device=dev))
               for _ in range(2)]
        ptrs = lambda t: [t[0:3].data_ptr(), t[3:5].data_ptr()]                    # shard 0 = pairs 0..2, shard 1 = pairs 3..4
        m.infer_device(n, ptrs(tx), out[0][0].data_ptr(), out[0][1].data_ptr())    # synchronous device form
        assert (out[0][0].cpu().numpy() == raw_x).all() and (out[0][1].cpu().numpy() == disp_x).all()
        # asynchronous, two batches in flight; a third is refused until one is waited for
        t1 = m.submit_device(n, ptrs(tx), out[0][0].data_ptr(), out[0][1].data_ptr())
        t2 = m.submit_device(n, ptrs(ty), out[1][0].data_ptr(), out[1][1].data_ptr())
        with pytest.raises(api.StereoNetError) as e:
            m.submit_device(n, ptrs(tx), out[0][0].data_ptr(), out[0][1].data_ptr())
        assert e.value.code == -6
        with pytest.raises(api.StereoNetError) as e:      # the host form shares the engines' workspaces: refused while
            m.infer(xs)                                   # device-resident batches are in flight (SN_ERR_BUSY)
        assert e.value.code == -6
        m.wait(t2)
        m.wait(t1)
        assert (out[0][0].cpu().numpy() == raw_x).all() and (out[1][0].cpu().numpy() == raw_y).all()
        assert (out[1][1].cpu().numpy() == disp_y).all()
        with pytest.raises(api.StereoNetError):
            m.wait(t1)
        # a missing shard is an error on every rank, not a hang
        with pytest.raises(api.StereoNetError):
            m.infer_device(n, [tx.data_ptr(), 0], out[0][0].data_ptr(), out[0][1].data_ptr())
        m.infer_device(n, ptrs(ty), out[0][0].data_ptr(), out[0][1].data_ptr())    # and the object still works
        assert (out[0][0].cpu().numpy() == raw_y).all()


@pytest.mark.gpu
def test_stream_priority_is_chosen_by_the_library(model_factory, monkeypatch):
    """One engine / one shard: default priority (the host-to-host paths are 30 % slower on high-priority streams, DESIGN.md
    §7); SN_STREAM_PRIORITY=1 / 0 force it either way, also against sn_mgpu_create's own choice."""
    monkeypatch.delenv("SN_STREAM_PRIORITY", raising=False)
    model = model_factory(96, 64, 48)
    with api.StereoNetHIP(model) as eng:
        assert eng.stream_priority_high() is False
    with api.StereoNetMultiGPU(model, devices=[0], max_batch=1) as m:
        assert m.engine_stream_priority_high(0) is False
    monkeypatch.setenv("SN_STREAM_PRIORITY", "1")
    with api.StereoNetHIP(model) as eng:
        assert eng.stream_priority_high() is True
    monkeypatch.setenv("SN_STREAM_PRIORITY", "0")
    monkeypatch.setenv("SN_MGPU_ALLOW_DUP", "1")
    with api.StereoNetMultiGPU(model, devices=[0, 0], max_batch=2) as m:
        assert m.engine_stream_priority_high(0) is False and m.engine_stream_priority_high(1) is False


@pytest.mark.gpu
def test_more_devices_than_present_is_an_error(model_factory):
    import torch
    have = torch.cuda.device_count()
    with pytest.raises(api.StereoNetError) as e:
        api.StereoNetMultiGPU(model_factory(96, 64, 48), ndev=have + 1, max_batch=have + 1)
    assert e.value.code == -1
    with pytest.raises(api.StereoNetError):
        api.StereoNetMultiGPU(model_factory(96, 64, 48), devices=[0, 0], max_batch=2)   # one shard per device
