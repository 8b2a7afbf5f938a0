"""plh_comm_* / plh_gather_records: the multi-GPU collection of the record blocks in the C ABI.
The N > 1 data path itself needs N GPUs (the driver's scaling run); here
  * the emulator build's one-rank communicator (CPU): argument checking, block layout, all-gather and gather-to-root forms,
  * on the GPU box a real one-rank RCCL communicator created from a unique id (ncclCommInitRank, grouped ncclAllGather /
    ncclGather on a stream) -- the same calls bench.py --gpus N makes with N ranks,
  * the sharding arithmetic with two gloo ranks (tests/test_distributed_cpu.py)."""
import numpy as np
import pytest


def _blocks(rng_seed, S):
    rng = S.SplitMix64(rng_seed)
    return [rng.randint(n, 0, 256).astype(np.uint8) for n in (4, 28 * 1006 * 3, 32 * 1006 * 3, 68 * 201 * 3)]


def test_emu_one_rank_gather(plslam, synth, emu_lib):
    P = plslam
    c = P.Comm(P.Comm.unique_id(emu_lib), 0, 1, lib=emu_lib)
    try:
        for root in (-1, 0):
            send = _blocks(5 + root, synth)
            recv = [np.zeros_like(b) for b in send]
            c.gather(list(zip(send, recv)), root=root)
            assert all((a == b).all() for a, b in zip(send, recv))
        with pytest.raises(P.PlhError):
            c.gather([(send[0], None)], root=-1)            # a receiving rank needs a receive buffer
        with pytest.raises(P.PlhError):
            c.gather([(send[0], recv[0])], root=3)          # root outside the communicator
    finally:
        c.close()
    with pytest.raises(P.PlhError):
        P.Comm(bytes(128), 1, 2, lib=emu_lib)               # the emulator has no RCCL: one rank only


@pytest.mark.gpu
def test_gpu_one_rank_rccl_gather(plslam, synth):
    import torch
    P = plslam
    P.load()
    c = P.Comm(P.Comm.unique_id(), 0, 1, device=0)
    try:
        assert c.rccl_version() > 20000                      # e.g. 22xxx: a real RCCL answered
        s = torch.cuda.Stream()
        for root in (-1, 0):
            send = [torch.from_numpy(b).cuda() for b in _blocks(9 + root, synth)]
            recv = [torch.zeros_like(b) for b in send]
            torch.cuda.synchronize()
            c.gather(list(zip(send, recv)), root=root, stream=s.cuda_stream)
            s.synchronize()
            assert all(bool((a == b).all()) for a, b in zip(send, recv))
    finally:
        c.close()
