"""cv2-free mask ingest used ONLY to build fixtures in this container.

Restates what reference utils/mask_utils.py:35-87 produces (PNG -> RGB; interior parts
recoloured to full_building before resizing; nearest-neighbour resize to max_dim with
dsize=(int(w*s), int(h*s)); binary = not background).  cv2 is absent here; nearest
sampling follows OpenCV's INTER_NEAREST rule src = min(floor(dst * src_n / dst_n), src_n-1).
SURVEY.md Appendix C: with this loader the reference reproduces its stored Taj grid.
"""
import os

import numpy as np
from PIL import Image

DATA = "/root/reference/data"


def nearest_resize(img, max_dim):
    h, w = img.shape[:2]
    s = max_dim / max(h, w)
    nw, nh = int(w * s), int(h * s)
    xs = np.minimum(np.floor(np.arange(nw) * (w / nw)).astype(int), w - 1)
    ys = np.minimum(np.floor(np.arange(nh) * (h / nh)).astype(int), h - 1)
    return np.ascontiguousarray(img[ys][:, xs])


def load_rgb(monument, view, suffix="_mask.png"):
    p = os.path.join(DATA, monument, "masks", f"{monument}_{view}{suffix}")
    return np.array(Image.open(p).convert("RGB"))


def load_and_prepare(monument, max_dim, part_colors, interior=("main_door", "windows")):
    sem = load_rgb(monument, "front")
    inter = np.zeros(sem.shape[:2], bool)
    for p in interior:
        inter |= np.all(sem == np.array(part_colors[p]), axis=-1)
    ext = sem.copy()
    ext[inter] = part_colors["full_building"]
    sem_r = nearest_resize(sem, max_dim)
    ext_r = nearest_resize(ext, max_dim)
    binary = (~np.all(ext_r == np.array(part_colors["background"]), axis=-1)).astype(np.uint8)
    return sem_r, ext_r, binary
