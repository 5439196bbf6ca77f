"""CPU restatement of the data-side callers of the path.  TEST INFRASTRUCTURE.

  * augment():  FlexibleDataset.__getitem__ + transforms of /root/reference/btsbot/train.py:178-199 and
    utils.py:44-48 for GIVEN random draws -- RandomHorizontalFlip = flip of the last (width) axis,
    RandomVerticalFlip = flip of the height axis, RandomRightAngleRotation = rotate(img, 0/90/180/270)
    counter-clockwise, which on a square cutout is torch.rot90(img, k, (-2, -1)).  torchvision is not
    installed here, so rot90's direction is pinned on a hand-written 3x3 case in the tests.
  * metrics():  val.py:159-168 / train.py:550-558 with torch's own BCEWithLogitsLoss.
"""
import torch


def augment(images: torch.Tensor, index, ops) -> torch.Tensor:
    """ops[b] = hflip | vflip << 1 | k << 2;  out[b] = rot90^k(vflip?(hflip?(images[index[b]])))."""
    n = len(ops) if ops is not None else (len(index) if index is not None else images.shape[0])
    out = []
    for b in range(n):
        img = images[int(index[b])] if index is not None else images[b]
        op = int(ops[b]) if ops is not None else 0
        if op & 1:
            img = torch.flip(img, dims=(-1,))
        if op & 2:
            img = torch.flip(img, dims=(-2,))
        img = torch.rot90(img, (op >> 2) & 3, dims=(-2, -1))
        out.append(img)
    return torch.stack(out) if out else images[:0]


def metrics(logits: torch.Tensor, labels: torch.Tensor, pos_weight: float):
    z = logits.reshape(-1, 1).float()
    y = labels.reshape(-1, 1).float()
    loss = torch.nn.BCEWithLogitsLoss(pos_weight=torch.tensor([pos_weight]))(z, y).item()
    acc = ((torch.sigmoid(z) > 0.5).float() == y).float().mean().item()
    return loss, acc


def make_triplet_arith(stamps, normalize: bool = True):
    """alert_utils.py:149-196 from the decoded stamps on (numpy, the reference's own calls): returns
    (triplet [63,63,3] float64, drop)."""
    import numpy as np
    import warnings
    cut = {}
    drop = False
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for name, data in zip(("science", "template", "difference"), stamps):
            data = np.array(data, dtype=np.float32)
            median = np.nanmedian(data.flatten())
            if median == np.nan or median == -np.inf or median == np.inf:
                drop = True
            cut[name] = np.nan_to_num(data)
            if normalize and not drop:
                cut[name] /= np.linalg.norm(cut[name])
            if np.all(cut[name].flatten() == 0):
                drop = True
            shape = cut[name].shape
            if shape != (63, 63):
                cut[name] = np.pad(cut[name], [(0, 63 - shape[0]), (0, 63 - shape[1])], mode="constant",
                                   constant_values=1e-9)
    triplet = np.zeros((63, 63, 3))
    triplet[:, :, 0] = cut["science"]
    triplet[:, :, 1] = cut["template"]
    triplet[:, :, 2] = cut["difference"]
    return triplet, drop


def alert_summary(raw_preds, labels):
    """val.py:178-218 on numpy with the reference's own calls (np.rint, bitwise masks, sklearn roc_curve + auc)."""
    import numpy as np
    from sklearn.metrics import roc_curve, auc
    raw_preds = np.asarray(raw_preds)
    preds = np.rint(raw_preds).astype(int)
    labels = np.asarray(labels).astype(int)
    fpr, tpr, _ = roc_curve(labels, raw_preds)
    roc_auc = auc(fpr, tpr)
    TP = int(np.bitwise_and(labels, preds).sum())
    TN = int((1 - np.bitwise_or(labels, preds)).sum())
    FP = int(np.bitwise_and(1 - labels, preds).sum())
    FN = int(np.bitwise_and(labels, 1 - preds).sum())
    bts_acc = TP / (TP + FN)
    notbts_acc = TN / (TN + FP)
    if TP > 0 and TN > 0:
        precision, recall = TP / (TP + FP), TP / (TP + FN)
    else:
        precision = recall = -999.0
    return {"roc_auc": float(roc_auc), "bal_acc": (bts_acc + notbts_acc) / 2, "bts_acc": bts_acc,
            "notbts_acc": notbts_acc, "alert_precision": precision, "alert_recall": recall,
            "TP": TP, "TN": TN, "FP": FP, "FN": FN}
