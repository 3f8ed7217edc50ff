"""Worker of tests/test_gpu_dp_procs.py: one rank of a two-process data-parallel job whose ranks share the box's single
GPU.  Collectives go through gloo on device tensors (RCCL refuses two ranks on one device); everything else - one engine
per process, the trainer's split forward / backward, the sync-BatchNorm all-reduces, the early gradient bucket on its
communication stream, the replica-state broadcast - is the path `bench.py --gpus N` runs.
argv: out_file model steps"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out, kind, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    if kind.startswith("nccl1:"):
        return nccl_one_rank(out, kind[6:], steps)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ecg_denoise_amd import NewRALE, RALENet, UNet
    from ecg_denoise_amd.dp import DataParallelTrainer, HipEngineAdapter, NewRALEEngineAdapter, UNetEngineAdapter
    Bg, L, leads = (16, 1024, 12) if kind == "newrale" else (128, 256, 2)
    g = torch.Generator().manual_seed(77)
    x = torch.randn(Bg, leads, L, generator=g); t = torch.randn(Bg, leads, L, generator=g)
    sh = Bg // world
    xl, tl = x[rank * sh:(rank + 1) * sh].cuda(), t[rank * sh:(rank + 1) * sh].cuda()
    # every rank draws DIFFERENT initial weights: the trainer must make rank 0's the job's
    if kind == "newrale":      # BASELINE config 4: 12 leads x 1024 samples, frozen inner RA-LENet + trainable adapter
        inner = RALENet("full", leads=2, L=L, max_batch=sh, device="cuda:0", seed=100 + rank)
        m = NewRALE(inner, seed=200 + rank)
        tr = DataParallelTrainer(NewRALEEngineAdapter(m))
    elif kind == "unet":
        m = UNet(leads=2, L=L, max_batch=sh, device="cuda:0", seed=100 + rank)
        tr = DataParallelTrainer(UNetEngineAdapter(m))
    else:
        m = RALENet("full", leads=2, L=L, max_batch=sh, device="cuda:0", seed=100 + rank)
        for k, v in m.named_parameters():
            if "relative_position_bias_table" in k:
                v.copy_(0.1 * torch.randn(v.shape, generator=torch.Generator().manual_seed(5 + rank)).cuda())
        tr = DataParallelTrainer(HipEngineAdapter(m))
    m.train()
    losses = []
    for _ in range(steps):
        losses.append(tr.train_step(xl, tl)["loss"].item())
    torch.cuda.synchronize()
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    grads = {k: v.cpu().clone() for k, v in m.named_grads().items()}      # the all-reduced (global-batch) gradient of the last step
    torch.save({"state": sd, "grads": grads, "losses": losses, "step_count": m.step_count}, f"{out}.rank{rank}")
    dist.barrier()
    dist.destroy_process_group()


def nccl_one_rank(out, kind, steps):
    """The REAL backend (RCCL, `nccl`) with the one rank a one-GPU box allows: process-group initialisation on the device, the
    trainer with `force_collectives` - BatchNorm all-reduces on the compute stream, the early gradient bucket on its
    communication stream behind `ral_grad_bucket_wait`'s events, the metric reduction - every one a true RCCL call (the
    identity over one rank), and the replica must end where the plain single-process step ends."""
    try:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
        probe = torch.ones(4, device="cuda:0")
        dist.all_reduce(probe)
        torch.cuda.synchronize()
    except Exception as exc:                      # no usable RCCL on this box: the caller skips
        torch.save({"skip": repr(exc)[:300]}, f"{out}.rank0")
        return
    from ecg_denoise_amd import RALENet, UNet
    from ecg_denoise_amd.dp import DataParallelTrainer, HipEngineAdapter, UNetEngineAdapter
    B, L = 128, 256
    g = torch.Generator().manual_seed(77)
    x = torch.randn(B, 2, L, generator=g).cuda(); t = torch.randn(B, 2, L, generator=g).cuda()
    if kind == "unet":
        m = UNet(leads=2, L=L, max_batch=B, device="cuda:0", seed=100)
        tr = DataParallelTrainer(UNetEngineAdapter(m), force_collectives=True)
    else:
        m = RALENet("full", leads=2, L=L, max_batch=B, device="cuda:0", seed=100)
        tr = DataParallelTrainer(HipEngineAdapter(m), force_collectives=True)
    m.train()
    losses, ncoll = [], []
    for _ in range(steps):
        losses.append(tr.train_step(x, t)["loss"].item())
        ncoll.append((tr.collectives_last_step, tr.metric_collectives_last_step))
    torch.cuda.synchronize()
    ver = None
    try:
        ver = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:
        pass
    torch.save({"state": {k: v.cpu() for k, v in m.state_dict().items()}, "losses": losses, "collectives": ncoll,
                "backend": dist.get_backend(), "rccl_version": ver, "step_count": m.step_count}, f"{out}.rank0")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
