"""Data loaders with the reference's entry point (src/data/data_loader.py:11).  With the parquet files present:
DataLoader workers decode the images and the reference's transform runs on the device for the whole batch (SURVEY 8f
row 2, src/data/transforms.py::BatchTransform).  When the parquet directory is absent: loaders over the synthetic
COCO-shaped dataset the benchmark uses, in the same (images, [targets]) format."""
import os

import torch
from torch.utils.data import DataLoader, Dataset
from torch.utils.data.distributed import DistributedSampler

from src.data.collate import collate_fn, raw_collate_fn


class DevicePreppedLoader:
    """A DataLoader of decoded images whose batches pass through the on-device transform on their way out: yields
    (images[N,3,S,S] on the device, [targets]) -- the format of src/data/collate.py."""

    def __init__(self, loader, transform):
        self.loader, self.transform = loader, transform
        self.sampler, self.dataset, self.batch_size = loader.sampler, loader.dataset, loader.batch_size

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        for images, targets in self.loader:
            yield self.transform(images, targets)


class DevicePrefetcher:
    """One batch of lookahead on a second HIP stream: while the training step of batch i runs, batch i+1 is produced on the
    side stream -- whatever device work the wrapped loader does to produce it (DevicePreppedLoader: the pinned upload and the
    transform kernels) plus the host -> device copy of a CPU image tensor (pinned by the DataLoader: asynchronous).  Copies
    and compute then overlap instead of queueing on one stream (a 32 x 3 x 640 x 640 fp32 batch is 157 MB: 3 - 6 ms of PCIe
    time against a 10.4 ms step).  The consumer's stream waits on the batch's event before it sees the tensor; the tensor is
    recorded on that stream so that the allocator does not hand its memory to the next upload while the step still reads it.
    Yields what the loader yields, with the images on the device."""

    def __init__(self, loader, device):
        self.loader, self.device = loader, torch.device(device)
        self.stream = None

    def __len__(self):
        return len(self.loader)

    def __getattr__(self, name):                # sampler / dataset / batch_size of the wrapped loader
        if name in ("loader", "device", "stream"):      # (not set yet: copy / pickle probing an empty instance)
            raise AttributeError(name)
        return getattr(self.loader, name)

    def _produce(self, it):
        with torch.cuda.stream(self.stream):
            try:
                images, targets = next(it)
            except StopIteration:
                return None
            if isinstance(images, torch.Tensor) and not images.is_cuda:
                images = images.to(self.device, non_blocking=True)
            ready = torch.cuda.Event()
            ready.record(self.stream)
        return images, targets, ready

    def __iter__(self):
        if self.device.type != "cuda":
            yield from self.loader
            return
        if self.stream is None:
            self.stream = torch.cuda.Stream(self.device)
        it = iter(self.loader)
        cur = self._produce(it)
        while cur is not None:
            images, targets, ready = cur
            consumer = torch.cuda.current_stream(self.device)
            consumer.wait_event(ready)
            if isinstance(images, torch.Tensor):
                images.record_stream(consumer)
            yield images, targets
            # the consumer has issued its step for this batch (asynchronously) and asks for the next one: produce it now, beside
            # that step
            cur = self._produce(it)


class SyntheticDetectionDataset(Dataset):
    """randn images; 1..20 boxes (cx,cy,w,h,cls) in pixels per image."""

    def __init__(self, length=64, res=640, num_classes=80, seed=1234):
        self.length, self.res, self.nc, self.seed = length, res, num_classes, seed

    def __len__(self):
        return self.length

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed + i)
        img = torch.randn(3, self.res, self.res, generator=g)
        m = int(torch.randint(1, 21, (1,), generator=g))
        boxes = torch.cat([torch.rand(m, 2, generator=g) * self.res, torch.rand(m, 2, generator=g) * (0.4 * self.res) + 8,
                           torch.randint(0, self.nc, (m, 1), generator=g).float()], 1)
        return img, {"boxes": boxes}


def get_data_loaders(train_parquet, val_parquet, train_images, val_images, batch_size, is_test=False, prefetch_factor=2,
                     percent=1.0, device="cpu", num_classes=80, res=640):
    if os.path.exists(train_parquet):
        # the reference's pipeline (src/data/data_loader.py:11-60): workers decode, the transform runs on the device
        from src.data.dataset_loader import DetectionDataset
        from src.data.transforms import get_train_transforms, get_val_transforms
        from src.utils.common import get_num_workers
        train_ds = DetectionDataset(train_parquet, train_images, is_test=is_test, percent=percent)
        val_ds = DetectionDataset(val_parquet, val_images, is_test=is_test, percent=percent)
        dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()
        ts = DistributedSampler(train_ds, shuffle=True) if dist_on else None
        vs = DistributedSampler(val_ds, shuffle=False) if dist_on else None
        nw = 0 if is_test else get_num_workers()
        kw = dict(num_workers=nw, collate_fn=raw_collate_fn, prefetch_factor=prefetch_factor if nw else None)
        train = DataLoader(train_ds, batch_size=batch_size, shuffle=ts is None, sampler=ts, drop_last=True, **kw)
        val = DataLoader(val_ds, batch_size=batch_size, shuffle=False, sampler=vs, **kw)
        return DevicePreppedLoader(train, get_train_transforms(res, device)), DevicePreppedLoader(val, get_val_transforms(res, device))
    n = 20 if is_test else 256
    train_ds = SyntheticDetectionDataset(max(batch_size, int(n * percent)), res, num_classes, 1234)
    val_ds = SyntheticDetectionDataset(max(batch_size, int(n * percent) // 4), res, num_classes, 4321)
    dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()
    ts = DistributedSampler(train_ds, shuffle=True) if dist_on else None
    vs = DistributedSampler(val_ds, shuffle=False) if dist_on else None
    pin = device == "cuda"
    train = DataLoader(train_ds, batch_size=batch_size, shuffle=ts is None, sampler=ts, num_workers=0, pin_memory=pin,
                       collate_fn=collate_fn, drop_last=True)
    val = DataLoader(val_ds, batch_size=batch_size, shuffle=False, sampler=vs, num_workers=0, pin_memory=pin,
                     collate_fn=collate_fn)
    return train, val
