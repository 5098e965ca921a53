"""PFM codec with the reference's call surface (datasets/data_io.py:6-71): `read_pfm(filename) -> (array, scale)`,
`save_pfm(filename, image, scale=1)`.  Depth and confidence maps travel between the MVS stage, the renderer and the
fusion filter in this format (runner.py:262-287,312-332).  Host-side byte shuffling only: rows are stored bottom-up,
the sign of the scale line carries the byte order."""
import re
import sys

import numpy as np


def read_pfm(filename):
    with open(filename, "rb") as f:
        magic = f.readline().decode("utf-8").rstrip()
        if magic not in ("PF", "Pf"):
            raise Exception("Not a PFM file.")
        channels = 3 if magic == "PF" else 1
        dims = re.match(r"^(\d+)\s(\d+)\s$", f.readline().decode("utf-8"))
        if not dims:
            raise Exception("Malformed PFM header.")
        width, height = (int(v) for v in dims.groups())
        scale = float(f.readline().rstrip())
        order = "<" if scale < 0 else ">"
        data = np.fromfile(f, order + "f")
    shape = (height, width, 3) if channels == 3 else (height, width)
    return np.flipud(np.reshape(data, shape)), abs(scale)


def save_pfm(filename, image, scale=1):
    image = np.asarray(image)
    if image.dtype.name != "float32":
        raise Exception("Image dtype must be float32.")
    if image.ndim == 3 and image.shape[2] == 3:
        magic = b"PF\n"
    elif image.ndim == 2 or (image.ndim == 3 and image.shape[2] == 1):
        magic = b"Pf\n"
    else:
        raise Exception("Image must have H x W x 3, H x W x 1 or H x W dimensions.")
    little = image.dtype.byteorder == "<" or (image.dtype.byteorder == "=" and sys.byteorder == "little")
    with open(filename, "wb") as f:
        f.write(magic)
        f.write(("%d %d\n" % (image.shape[1], image.shape[0])).encode("utf-8"))
        f.write(("%f\n" % (-scale if little else scale)).encode("utf-8"))
        np.flipud(image).tofile(f)
