"""NumPy restatement of the seeded synthetic inputs for tests -- independent of
the device generators in csrc/synth.hip, which are checked against it."""
import numpy as np

BASE10 = np.array([(216, 224, 251), (253, 248, 96), (1, 220, 5), (63, 138, 173), (190, 0, 255), (0, 0, 255),
                   (5, 223, 223), (255, 180, 80), (180, 140, 255), (255, 120, 230)], np.uint8)


def palette16():
    pal = np.zeros((16, 3), np.uint8)
    pal[:10] = BASE10
    for k in range(10, 16):
        pal[k] = (16 * k, 255 - 16 * k, 8 * k + 7)
    return pal


def mask16(S):
    y, x = np.mgrid[0:S, 0:S].astype(np.int64)
    xn, yn = (x * 1024) // S, (y * 1024) // S
    dx2 = 2 * xn - 1023
    adx2 = np.abs(dx2)
    body = (adx2 < 840) & (yn >= 256)
    dy2 = 2 * (yn - 256)
    dome = dx2 * dx2 * 40000 + dy2 * dy2 * 90000 < 4 * 90000 * 40000
    towers = (adx2 > 880) & (adx2 < 960) & (yn >= 96)
    fg = body | dome | towers
    lab = np.where(fg, 1 + ((xn >> 6) + 3 * (yn >> 7)) % 15, 0).astype(np.uint8)
    return lab, (lab != 0).astype(np.uint8), palette16()[lab]


def splitmix64(z):
    z = np.asarray(z, np.uint64)
    with np.errstate(over="ignore"):
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def sem_slab(x0, x1, H, D, seed):
    idx = np.arange(x0 * H * D, x1 * H * D, dtype=np.uint64)
    lab = (splitmix64(np.uint64(seed) ^ idx) & np.uint64(15)).astype(np.int64)
    return palette16()[lab].reshape(x1 - x0, H, D, 3)


def occ_slab(x0, x1, H, D, seed):
    idx = np.arange(x0 * H * D, x1 * H * D, dtype=np.uint64)
    return (splitmix64(np.uint64(seed) ^ idx) & np.uint64(1)).astype(np.uint8).reshape(x1 - x0, H, D)
