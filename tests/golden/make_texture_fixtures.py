"""Writes extra small texture files (PNG / TGA / BMP / PGM variants) under tests/golden/obj/textures with PIL.
The expected RGBA8 pixels are NOT produced here: `make -C oracle fixtures` decodes every file with the
reference's own decoder (external/stb/stb_image.h, as Model.cpp:350 calls it) into host_golden.json."""
import os
import struct

import numpy as np
from PIL import Image

D = os.path.join(os.path.dirname(os.path.abspath(__file__)), "obj", "textures")
rng = np.random.default_rng(42)


def rnd(h, w, c):
    return rng.integers(0, 256, (h, w, c), dtype=np.uint8)



def write_adam7(path, img):
    """PIL cannot write interlaced PNG: assemble one by hand (8-bit RGB, filter type 0 on every row)."""
    import zlib

    h, w, _ = img.shape
    raw = b""
    for x0, y0, dx, dy in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
        sub = img[y0::dy, x0::dx]
        if sub.shape[0] == 0 or sub.shape[1] == 0:
            continue
        for row in sub:
            raw += b"\x00" + row.tobytes()

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 1))
                + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


write_adam7(os.path.join(D, "x_adam7.png"), rnd(11, 13, 3))
write_adam7(os.path.join(D, "x_adam7_small.png"), rnd(2, 3, 3))
Image.fromarray(rnd(5, 11, 4), "RGBA").save(os.path.join(D, "x_rgba.png"), compress_level=9)
Image.fromarray(rnd(6, 7, 2)[..., 0], "L").convert("LA").save(os.path.join(D, "x_la.png"))
Image.fromarray(rnd(7, 9, 1)[..., 0], "L").convert("P").save(os.path.join(D, "x_pal.png"))
p = Image.fromarray((rnd(6, 10, 1)[..., 0] % 7).astype(np.uint8), "P")
p.putpalette([int(v) for v in rnd(1, 7, 3).reshape(-1)])
p.save(os.path.join(D, "x_pal_trns.png"), transparency=bytes([0, 128, 255, 17]), bits=4)
Image.fromarray((rnd(5, 9, 1)[..., 0] > 127).astype(np.uint8) * 255, "L").convert("1").save(os.path.join(D, "x_bw1.png"))
Image.fromarray(rng.integers(0, 65536, (4, 6), dtype=np.uint16).astype(np.int32), "I").convert("I;16").save(os.path.join(D, "x_gray16.png"))
Image.fromarray(rnd(4, 5, 3), "RGB").save(os.path.join(D, "x_rgb_trns.png"), transparency=(12, 200, 7))
Image.fromarray(rnd(40, 50, 3) // 64 * 64, "RGB").save(os.path.join(D, "x_rgb_big.png"), compress_level=1)
Image.fromarray(rnd(37, 23, 3), "RGB").save(os.path.join(D, "x_rgb_stored.png"), compress_level=0)
# TGA / BMP / PGM
Image.fromarray(rnd(5, 6, 4), "RGBA").save(os.path.join(D, "x_rgba.tga"))
Image.fromarray(rnd(5, 6, 3), "RGB").save(os.path.join(D, "x_rgb_rle.tga"), compression="tga_rle")
Image.fromarray(rnd(4, 7, 1)[..., 0], "L").save(os.path.join(D, "x_gray.tga"))
Image.fromarray(rnd(4, 7, 1)[..., 0], "L").convert("P").save(os.path.join(D, "x_pal.tga"))
Image.fromarray(rnd(5, 9, 3), "RGB").save(os.path.join(D, "x_rgb24.bmp"))
Image.fromarray(rnd(5, 9, 1)[..., 0], "L").convert("P").save(os.path.join(D, "x_pal8.bmp"))
Image.fromarray(rnd(5, 9, 4), "RGBA").save(os.path.join(D, "x_rgba32.bmp"))
Image.fromarray(rnd(6, 5, 1)[..., 0], "L").save(os.path.join(D, "x_gray.pgm"))
# 16-bit (5-5-5) top-down TGA written by hand
w, h = 5, 3
px = rng.integers(0, 32768, (h, w), dtype=np.uint16)
with open(os.path.join(D, "x_rgb16.tga"), "wb") as f:
    f.write(struct.pack("<BBBHHBHHHHBB", 0, 0, 2, 0, 0, 0, 0, 0, w, h, 16, 0x20))
    f.write(px.astype("<u2").tobytes())
print("written to", D)
