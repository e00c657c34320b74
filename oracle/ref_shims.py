"""oracle/ref_shims.py — TEST INFRASTRUCTURE ONLY (build container only).

Makes the read-only reference at /root/reference importable in this container so that
oracle/make_golden.py can run the reference's OWN functions and record their outputs as
golden vectors.  Nothing from /root/reference is copied; this module only installs stand-in
*third-party* modules (clip, open_clip, cv2, torchvision, torchmetrics — packages the
reference imports at module top level but that are absent from this image) and patches two
torch behaviours.  It is never imported on the GPU box.
"""
import os
import sys
import types

import torch

REF = "/root/reference"


def install():
    if getattr(install, "_done", False):
        return
    if not os.path.isdir(REF):
        raise RuntimeError("reference tree not present: golden generation only runs in the build container")

    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    mod("clip")
    mod("open_clip")
    mod("cv2")

    class _T:  # placeholder transform classes, imported by name at utils/extras.py:2
        def __init__(self, *a, **k):
            pass

    names = ["Compose", "Resize", "CenterCrop", "ToTensor", "Normalize", "RandomResizedCrop",
             "RandomHorizontalFlip", "RandomCrop"]
    tv = mod("torchvision")
    tvt = mod("torchvision.transforms", **{n: type(n, (_T,), {}) for n in names})
    tv.transforms = tvt
    tvd = mod("torchvision.datasets", Flowers102=type("Flowers102", (_T,), {}))
    tvf = mod("torchvision.datasets.folder", default_loader=lambda p: None)
    tvd.folder = tvf
    tv.datasets = tvd

    class ConfusionMatrix:  # torchmetrics stand-in: counts m[target, pred] (testing.py:417,458,483)
        def __init__(self, task="multiclass", num_classes=2):
            self.m = torch.zeros(num_classes, num_classes, dtype=torch.int64)

        def update(self, preds, target):
            for p, t in zip(preds.tolist(), target.tolist()):
                self.m[t, p] += 1

        def compute(self):
            return self.m

    mod("torchmetrics", ConfusionMatrix=ConfusionMatrix)

    # no GPU in the build container: .cuda() is the identity (utils/training.py:1462, testing.py:427,442)
    torch.Tensor.cuda = lambda self, *a, **k: self

    # utils/optimizers.py:170 passes `verbose` positionally; torch>=2.7 dropped the argument.
    from torch.optim import lr_scheduler
    _orig = lr_scheduler.LRScheduler.__init__

    def _init(self, optimizer, last_epoch=-1, *ignored, **kw):
        kw.pop("verbose", None)
        _orig(self, optimizer, last_epoch)

    lr_scheduler.LRScheduler.__init__ = _init

    sys.path.insert(0, REF)
    install._done = True


def import_retrieval():
    """retrieval/sample_retrieval.py opens '../config.yml' relative to cwd at import time."""
    install()
    cwd = os.getcwd()
    os.chdir(os.path.join(REF, "retrieval"))
    sys.path.insert(0, os.path.join(REF, "retrieval"))
    try:
        import sample_retrieval  # noqa
    finally:
        os.chdir(cwd)
    return sample_retrieval
