"""Downstream scoring of denoised windows (reference test_cls.py:14-29): accuracy, precision and F1 of a two-class
beat classifier's logits (N vs V beats, label 1 = V).  The classifier itself (`model/ResNet_cls.py`) is built on
`global_utils.torch_utils.layers.Bottleneck1d`, an un-vendored dependency whose source is not in the reference
repository, so it cannot be restated here; these three functions take the logits of whatever classifier the caller has
and reproduce the reference's arithmetic (argmax over dim 1, counts as Python floats, no zero-division guard)."""
import torch


def acc(pred, label):
    p = torch.argmax(pred, dim=1)
    return torch.sum(p == label).item() / len(label)


def precision(pred, label):
    p = torch.argmax(pred, dim=1)
    tp = torch.sum(p * label).item()
    fp = torch.sum(p * (1 - label)).item()
    return tp / (tp + fp)


def f1_score(pred, label):
    p = torch.argmax(pred, dim=1)
    tp = torch.sum(p * label).item()
    fp = torch.sum(p * (1 - label)).item()
    fn = torch.sum((1 - p) * label).item()
    return tp / (tp + 0.5 * (fp + fn))


def score_denoiser(classifier, denoiser, loader, device):
    """The per-model block of test_cls.py:152-255: classify `denoiser(data)` (or the raw data when `denoiser` is None)
    for every batch of `loader` and return (acc, precision, f1) over the whole set."""
    preds, labels = [], []
    with torch.no_grad():
        for data, label in loader:
            data = torch.as_tensor(data, dtype=torch.float32).to(device)
            if denoiser is not None:
                data = denoiser(data.contiguous())
            preds.append(classifier(data)); labels.append(torch.as_tensor(label).long().to(device))
    p, l = torch.cat(preds), torch.cat(labels)
    return acc(p, l), precision(p, l), f1_score(p, l)
