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
def test_more_devices_than_present_is_an_error(model_factory):
    import torch
    have = torch.cuda.device_count()
    with pytest.raises(api.StereoNetError) as e:
        api.StereoNetMultiGPU(model_factory(96, 64, 48), ndev=have + 1, max_batch=have + 1)
    assert e.value.code == -1
    with pytest.raises(api.StereoNetError):
        api.StereoNetMultiGPU(model_factory(96, 64, 48), devices=[0, 0], max_batch=2)   # one shard per device
