"""The reference's DetectionDataset (src/data/dataset_loader.py:14-82): one parquet row per image (file_name, bbox list in
XYWH pixels, category_id list, name).  Here an item is the DECODED image (uint8 H x W x 3) and its untransformed target --
the transform runs on the device for the whole batch (src/data/transforms.py::BatchTransform)."""
import os

import numpy as np
import pandas as pd
import torch
from PIL import Image
from torch.utils.data import Dataset


class DetectionDataset(Dataset):
    def __init__(self, parquet_path, image_dir, transform=None, is_test=False, percent: float = 1.0):
        self.df = pd.read_parquet(parquet_path)
        self.df = self.df.sample(frac=percent)
        print("[INFO] Using {:0.2f}% of the dataset".format(percent * 100))
        print("[INFO] Loaded parquet file - {}".format(parquet_path))
        if is_test:
            self.df = self.df.head(20)
            print("[INFO] Reducing data for test")
        self.image_dir = image_dir
        self.transform = transform          # kept for signature parity; the batch transform is applied by the loader

    def __len__(self):
        return len(self.df)

    def __getitem__(self, idx):
        row = self.df.iloc[idx]
        image = np.array(Image.open(os.path.join(self.image_dir, row["file_name"])).convert("RGB"))      # a writable copy
        boxes = torch.from_numpy(np.array([list(b) for b in row["bbox"]], dtype=np.float32).reshape(-1, 4))
        labels = torch.from_numpy(np.array(list(row["category_id"]), dtype=np.float32)).reshape(-1, 1)
        return torch.from_numpy(np.ascontiguousarray(image)), {"boxes": boxes, "labels": labels, "image_id": torch.tensor([idx]),
                                                                "name": row["name"]}
