"""2-D mask ingest on the path's edge (host side; images are at most a few megapixels) -- row N3.

Mirrors reference utils/mask_utils.py: load_mask (:14-33), load_and_prepare_masks (:35-87) and
mask_parts_from_image (:89-97).  Upstream decodes and resizes with OpenCV, which this build does not depend on:
PNGs are decoded with Pillow and resized with OpenCV's INTER_NEAREST rule
    src = min(floor(dst * src_n / dst_n), src_n - 1),   dsize = (int(w*s), int(h*s)),  s = max_dim / max(h, w)
(with this rule the pipeline reproduces the reference's stored Taj grid position-exactly).

Parity note (unpinned): upstream calls cv2.resize(img, dsize, cv2.INTER_NEAREST) (mask_utils.py:60) -- the third POSITIONAL slot of
cv2.resize is `dst`, not `interpolation`; a cv2 build that accepts the integer there resizes with its default INTER_LINEAR.  cv2 is
not available in this image, so which one upstream's environment ran cannot be observed; the stored results/1 artefacts are consistent
with nearest, which is what this module implements (`interpolation="nearest"`).  `interpolation="linear"` is offered for callers who
know their cv2 took the default: OpenCV's half-pixel bilinear rule in fixed point is NOT reproduced bit for bit -- it is a plain
float bilinear resize and is outside every parity claim."""
import os

import numpy as np

__all__ = ["load_mask", "load_and_prepare_masks", "mask_parts_from_image", "resize_to_max"]


def _read_rgb(path):
    from PIL import Image
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    return np.array(Image.open(path).convert("RGB"))


def resize_to_max(img, max_dim, interpolation="nearest"):
    """resize so that the longer side becomes max_dim (reference :57-60); see the module docstring for `interpolation`"""
    h, w = img.shape[:2]
    if interpolation == "linear":
        from PIL import Image
        s = max_dim / max(h, w)
        return np.array(Image.fromarray(np.asarray(img)).resize((int(w * s), int(h * s)), Image.BILINEAR))
    if interpolation != "nearest":
        raise ValueError("interpolation is 'nearest' or 'linear'")
    s = max_dim / max(h, w)
    nw, nh = int(w * s), int(h * s)
    if nw <= 0 or nh <= 0:
        raise ValueError("resize to an empty image")
    # OpenCV's INTER_NEAREST index rule, with its own arithmetic (imgproc/resize.cpp): inv_scale = (double)dst / src;
    # ifx = 1.0 / inv_scale; sx = min(cvFloor(x * ifx), src - 1).  (x * (src / dst) differs from it by an ulp for some size pairs.)
    ifx = 1.0 / (nw / float(w)); ify = 1.0 / (nh / float(h))
    xs = np.minimum(np.floor(np.arange(nw) * ifx).astype(np.int64), w - 1)
    ys = np.minimum(np.floor(np.arange(nh) * ify).astype(np.int64), h - 1)
    return np.ascontiguousarray(img[ys][:, xs])


def load_mask(root_path, monument_name, view_name, max_dim=None):
    path = os.path.join(root_path, monument_name, "masks", f"{monument_name}_{view_name}_mask.png")
    mask = _read_rgb(path)
    if max_dim is not None:
        mask = resize_to_max(mask, max_dim)
    return mask


def load_and_prepare_masks(root_path, monument_name, view_name, max_dim, part_colors_np, interior_parts, visualize=False):
    """(semantic mask, exterior mask with interior parts painted as full_building, binary carving mask), all resized
    to max_dim; the recolouring happens BEFORE the resize, as upstream (:48-54)."""
    mask_dir = os.path.join(root_path, monument_name, "masks")
    semantic = _read_rgb(os.path.join(mask_dir, f"{monument_name}_{view_name}_mask.png"))
    interior = np.zeros(semantic.shape[:2], bool)
    for part in interior_parts:
        interior |= np.all(semantic == part_colors_np[part], axis=-1)
    exterior = semantic.copy()
    exterior[interior] = part_colors_np["full_building"]
    semantic_r = resize_to_max(semantic, max_dim)
    exterior_r = resize_to_max(exterior, max_dim)
    if monument_name == "Charminar":   # visualisation-only override upstream (:66-71)
        win = os.path.join(mask_dir, f"{monument_name}_{view_name}_mask_win.png")
        if os.path.exists(win):
            semantic_r = resize_to_max(_read_rgb(win), max_dim)
    binary = (~np.all(exterior_r == part_colors_np["background"], axis=-1)).astype(np.uint8)
    if visualize:
        import matplotlib.pyplot as plt
        fig, axs = plt.subplots(1, 3, figsize=(12, 4))
        for ax, im, title in zip(axs, (semantic_r, exterior_r, binary), ("Original Mask", "Exterior Mask", "Binary Mask")):
            ax.imshow(im, cmap="gray" if im.ndim == 2 else None); ax.set_title(title); ax.axis("off")
        plt.tight_layout(); plt.show()
    return semantic_r, exterior_r, binary


def mask_parts_from_image(image, part_colors, selected_parts):
    image = np.asarray(image)
    mask = np.zeros_like(image)
    for part in selected_parts:
        color = part_colors[part]
        match = np.all(image == color, axis=-1)
        mask[match] = color
    return mask
