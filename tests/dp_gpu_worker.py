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


if __name__ == "__main__":
    main()
