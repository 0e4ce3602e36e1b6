"""Headless frames of the coverage task: a numpy rasteriser standing in for the reference's pyglet viewer
(uav_dcc_control/envs/mpe/multiagent/rendering.py + environment.py:209-290, which need a display).

What the reference's viewer shows is kept: the PoIs as small dots whose colour follows their accumulated energy (grey ->
orange, green once covered), the UAVs as discs with their coverage radius drawn around them, and the communication links
between UAVs within range of each other.  One frame is a [size, size, 3] uint8 array, so `render("rgb_array")[0][0]`
(reference learner.py:199-200) is an image that an animated GIF can be assembled from.
"""
import numpy as np

EXTENT = 1.6            # world half-width shown (the task ends when a UAV leaves |x| <= 1.5)
BG, GRID = (255, 255, 255), (232, 232, 232)
UAV, UAV_RING, LINK = (40, 90, 200), (150, 180, 235), (90, 90, 90)
POI_0, POI_1, POI_DONE = np.array((170, 170, 170.0)), np.array((240, 140, 30.0)), (40, 170, 70)


def _px(v, size):
    return (np.asarray(v, np.float64) + EXTENT) * (size / (2 * EXTENT))


def _disc(img, cx, cy, r, color, ring=False):
    n = img.shape[0]
    x0, x1 = max(0, int(cx - r - 2)), min(n, int(cx + r + 3))
    y0, y1 = max(0, int(cy - r - 2)), min(n, int(cy + r + 3))
    if x0 >= x1 or y0 >= y1:
        return
    yy, xx = np.mgrid[y0:y1, x0:x1]
    d = np.hypot(xx + 0.5 - cx, yy + 0.5 - cy)
    m = (np.abs(d - r) <= 0.8) if ring else (d <= r)
    img[y0:y1, x0:x1][m] = color


def _line(img, a, b, color):
    n = int(max(abs(b[0] - a[0]), abs(b[1] - a[1]))) + 1
    xs = np.clip(np.linspace(a[0], b[0], n).astype(int), 0, img.shape[1] - 1)
    ys = np.clip(np.linspace(a[1], b[1], n).astype(int), 0, img.shape[0] - 1)
    img[ys, xs] = color


def rasterize(pos, poi, energy, done, r_cover, m_energy, r_comm=None, size=350):
    """One env: pos [N,2], poi [M,2], energy [M], done [M] -> uint8 [size, size, 3] (y up, like the viewer)."""
    img = np.empty((size, size, 3), np.uint8)
    img[:] = BG
    for g in (-1.0, 0.0, 1.0):                           # the unit box and the axes
        c = int(_px(g, size))
        if 0 <= c < size:
            img[:, c] = GRID; img[c, :] = GRID
    scale = size / (2 * EXTENT)
    pp, up = _px(poi, size), _px(pos, size)
    frac = np.clip(np.asarray(energy, np.float64) / float(m_energy), 0.0, 1.0)
    for j in range(len(pp)):
        col = POI_DONE if done[j] else tuple((POI_0 + (POI_1 - POI_0) * frac[j]).astype(np.uint8))
        _disc(img, pp[j, 0], pp[j, 1], max(2.0, 0.02 * scale), col)
    if r_comm:
        for a in range(len(up)):
            for b in range(a + 1, len(up)):
                if np.hypot(*(np.asarray(pos[a]) - np.asarray(pos[b]))) < 2 * r_comm:
                    _line(img, up[a], up[b], LINK)
    for i in range(len(up)):
        _disc(img, up[i, 0], up[i, 1], r_cover * scale, UAV_RING, ring=True)
        _disc(img, up[i, 0], up[i, 1], max(2.5, 0.03 * scale), UAV)
    return img[::-1].copy()


def save_gif(frames, path, duration_s=0.1):
    """Animated GIF of a list of frames (the reference writes one per render rollout with imageio, learner.py:204-210;
    imageio is not a dependency here, Pillow does it)."""
    from PIL import Image
    ims = [Image.fromarray(f) for f in frames]
    ims[0].save(path, save_all=True, append_images=ims[1:], duration=int(duration_s * 1000), loop=0)
