"""Downstream scoring of denoised windows (reference test_cls.py:14-29): accuracy, precision and F1 of a two-class
beat classifier's logits (N vs V beats, label 1 = V).  The classifier itself (`model/ResNet_cls.py`) is built on
`global_utils.torch_utils.layers.Bottleneck1d`, an un-vendored dependency whose source is not in the reference
repository, so it cannot be restated here; these three functions take the logits of whatever classifier the caller has
and reproduce the reference's arithmetic (argmax over dim 1, counts as Python floats, no zero-division guard)."""
import torch


def confusion(pred, label):
    """(tp, fp, fn, tn) of the argmax decision against binary labels (1 = V beat), counted in one pass: the four cells are
    the histogram of 2 * label + decision."""
    d = torch.argmax(pred, dim=1).long()
    cells = torch.bincount(2 * label.long().reshape(-1) + d.reshape(-1), minlength=4).tolist()
    tn, fp, fn, tp = (float(c) for c in cells[:4])
    return tp, fp, fn, tn


def acc(pred, label):
    tp, fp, fn, tn = confusion(pred, label)
    return (tp + tn) / len(label)


def precision(pred, label):
    tp, fp, _, _ = confusion(pred, label)
    return tp / (tp + fp)              # (no zero-division guard, as the reference: a classifier that never says V raises)


def f1_score(pred, label):
    tp, fp, fn, _ = confusion(pred, label)
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
    tp, fp, fn, tn = confusion(torch.cat(preds), torch.cat(labels))
    return (tp + tn) / (tp + fp + fn + tn), tp / (tp + fp), tp / (tp + 0.5 * (fp + fn))
