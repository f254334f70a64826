"""2-D mask helpers on the path's edge (host side; images are at most a few megapixels).

mask_parts_from_image mirrors reference utils/mask_utils.py:89-97; it produces the part image the camera
objective compares projections with (reference utils/camera_estimation.py:489)."""
import numpy as np

__all__ = ["mask_parts_from_image"]


def mask_parts_from_image(image, part_colors, selected_parts):
    image = np.asarray(image)
    mask = np.zeros_like(image)
    for part in selected_parts:
        color = part_colors[part]
        match = np.all(image == color, axis=-1)
        mask[match] = color
    return mask
