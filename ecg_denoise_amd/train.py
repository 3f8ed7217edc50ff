"""Host-side counterpart of the reference training harness, denoise_train.py::train
(reference lines in brackets).  Same protocol, same return value, same files:

  * Adam(lr=1e-3) + F.mse_loss mean                               [denoise_train.py:24,53]
  * per-step per-window SNR / RMSE on the TRAIN-mode outputs        [:58-59]
  * eval pass each epoch with BatchNorm running statistics          [:66-81]
  * epoch value = mean over windows                                 [:82-89]
  * state_dict saved every 10th epoch as model_save/{name}/{name}_{epoch}_{noise}_intensity{n}.pth  [:90-93]
  * one line appended to ./output.txt                               [:100-101]

Differences on purpose: metrics come out of the fused loss kernel ((B,) tensors, no autograd graph
kept alive — reference quirk A10), the progress-bar loss is not divided by the batch size (A12), and
`model_path` really resumes (the reference rebinds `model` to load_state_dict's return value, A11).
"""
import os

import torch


def _to_dev(t, device):
    t = torch.as_tensor(t, dtype=torch.float32)
    return t.to(device, non_blocking=True).contiguous()


def train(epochs, model, batch_size, train_loader, test_loader, use_gpu=True, model_path=None, *args, **kwargs):
    model_name = kwargs["model_name"]
    noise_name, noise_intensity = kwargs["noise_name"], kwargs["noise_intensity"]
    out_dir = kwargs.get("out_dir", ".")
    log = kwargs.get("log", print)
    lr = kwargs.get("lr", 1e-3)
    if not use_gpu:
        raise RuntimeError("the RA-LENet path runs on a HIP device only (no CPU fallback)")
    device = model.eng.device if hasattr(model, "eng") else model.device      # NewRALE wraps an engine-backed model
    if model_path:
        model.load_state_dict(torch.load(model_path, map_location="cpu"))
    train_snr_list, test_snr_list, train_rmse_list, test_rmse_list = [], [], [], []
    train_loss_list, eval_loss_list = [], []
    for epoch in range(epochs):
        model.train()
        snr_acc, rmse_acc, losses = [], [], []
        for data, target in train_loader:
            data, target = _to_dev(data, device), _to_dev(target, device)
            out = model.train_step(data, target, lr)
            losses.append(out["loss"])
            snr_acc.append(out["snr"]); rmse_acc.append(out["rmse"])
        model.eval()
        tsnr, trmse, elosses = [], [], []
        for data, target in test_loader:
            data, target = _to_dev(data, device), _to_dev(target, device)
            pred = model(data)
            loss, snr, rmse = model.loss_and_metrics(pred, target, want_grad=False)
            elosses.append(loss); tsnr.append(snr); trmse.append(rmse)
        nanf = torch.full((1,), float("nan"), device=device)
        if not snr_acc:   # evaluation-only call (no training batches)
            snr_acc, rmse_acc, losses = [nanf], [nanf], [nanf.double()]
        train_snr = torch.cat(snr_acc).mean().item(); train_rmse = torch.cat(rmse_acc).mean().item()
        test_snr = torch.cat(tsnr).mean().item(); test_rmse = torch.cat(trmse).mean().item()
        train_snr_list.append(train_snr); test_snr_list.append(test_snr)
        train_rmse_list.append(train_rmse); test_rmse_list.append(test_rmse)
        train_loss_list.append(torch.cat(losses).mean().item()); eval_loss_list.append(torch.cat(elosses).mean().item())
        if (epoch + 1) % 10 == 0:
            d = os.path.join(out_dir, "model_save", model_name)
            os.makedirs(d, exist_ok=True)
            path = os.path.join(d, f"{model_name}_{epoch}_{noise_name}_intensity{noise_intensity}.pth")
            torch.save(model.state_dict(), path)
            log(f"{path}\nepoch: {epoch + 1}\ntrain snr: {train_snr}\ntest snr: {test_snr}\n"
                f"train rmse: {train_rmse}\ntest rmse: {test_rmse}")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "output.txt"), "a") as f:
        f.write(f"{model_name}_{epoch}_{noise_name}_intensity{noise_intensity}:snr:{test_snr}, rmse:{test_rmse}\n")
    train.last_losses = (train_loss_list, eval_loss_list)
    return train_snr_list, test_snr_list, train_rmse_list, test_rmse_list

