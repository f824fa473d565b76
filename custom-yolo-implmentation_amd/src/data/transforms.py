"""Validation transform used by Model.inference on PIL input (reference: src/data/transforms.py:16-24):
resize to 640x640, scale to [0,1], ImageNet normalisation.  torchvision is not required: PIL + torch."""
import numpy as np
import torch

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def get_val_transforms(size=640):
    def apply(img):
        from PIL import Image
        img = img.convert("RGB").resize((size, size), Image.BILINEAR)
        x = torch.from_numpy(np.asarray(img, dtype=np.float32) / 255.0).permute(2, 0, 1)
        return (x - torch.tensor(MEAN).view(3, 1, 1)) / torch.tensor(STD).view(3, 1, 1)
    return apply


def get_train_transforms(size=640):
    """Training-time augmentation lives in the (out-of-scope) data pipeline; geometry-free part only."""
    return get_val_transforms(size)
