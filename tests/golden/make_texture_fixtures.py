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

# ---- JPEG (libjpeg through PIL): baseline / progressive, every chroma layout PIL offers, grey, RGB, CMYK, restart
# intervals, optimised tables, plus hand-edited streams (fill bytes, comment, early end of the entropy data)
jrng = np.random.default_rng(77)


def smooth(h, w, c=3, noise=24):
    """low-frequency content plus noise: realistic coefficient statistics (runs, EOBs) at small sizes"""
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([128 + 100 * np.sin(xx / (3.0 + k) + k) * np.cos(yy / (2.0 + k)) for k in range(c)], -1)
    img = base + jrng.normal(0, noise, (h, w, c))
    return np.clip(img, 0, 255).astype(np.uint8)


def jsave(name, arr, mode="RGB", **kw):
    a = arr if mode != "L" else arr[..., 0]
    Image.fromarray(a, mode).save(os.path.join(D, name), "JPEG", **kw)


jsave("j_base_444.jpg", smooth(13, 19), quality=85, subsampling=0)
jsave("j_base_422.jpg", smooth(9, 21), quality=80, subsampling=1)
jsave("j_base_420.jpg", smooth(17, 33), quality=75, subsampling=2)
jsave("j_base_420_wide.jpg", smooth(20, 70), quality=90, subsampling=2)
jsave("j_base_411.jpg", smooth(11, 37), quality=80, subsampling="4:1:1")
jsave("j_gray.jpg", smooth(10, 15, 1), mode="L", quality=70)
jsave("j_1x1_420.jpg", smooth(1, 1), quality=90, subsampling=2)
jsave("j_2x2_420.jpg", smooth(2, 2), quality=90, subsampling=2)
jsave("j_1x9_422.jpg", smooth(9, 1), quality=90, subsampling=1)
jsave("j_noise_q100.jpg", jrng.integers(0, 256, (16, 16, 3), dtype=np.uint8), quality=100, subsampling=0)
jsave("j_noise_q5.jpg", jrng.integers(0, 256, (24, 24, 3), dtype=np.uint8), quality=5, subsampling=2)
jsave("j_optimize.jpg", smooth(20, 28), quality=95, subsampling=2, optimize=True)
jsave("j_restart_420.jpg", smooth(33, 41), quality=80, subsampling=2, restart_marker_blocks=2)
jsave("j_restart_rows_444.jpg", smooth(25, 18), quality=80, subsampling=0, restart_marker_rows=1)
jsave("j_prog_420.jpg", smooth(24, 40), quality=80, subsampling=2, progressive=True)
jsave("j_prog_444_q30.jpg", smooth(32, 48, noise=6), quality=30, subsampling=0, progressive=True)
jsave("j_prog_q98.jpg", smooth(19, 23, noise=40), quality=98, subsampling=1, progressive=True)
jsave("j_prog_gray.jpg", smooth(21, 17, 1), mode="L", quality=85, progressive=True)
jsave("j_prog_restart.jpg", smooth(26, 34), quality=70, subsampling=2, progressive=True, restart_marker_blocks=3)
jsave("j_rgb_keep.jpg", smooth(14, 22), quality=90, keep_rgb=True)
Image.fromarray(smooth(12, 20, 4), "CMYK").save(os.path.join(D, "j_cmyk.jpg"), "JPEG", quality=85)
Image.fromarray(smooth(18, 26, 4), "CMYK").save(os.path.join(D, "j_cmyk_prog.jpg"), "JPEG", quality=85, progressive=True)


def markers(b):
    """offsets of the marker segments up to and including SOS: [(offset, marker, length incl. the 2-byte size)]"""
    out, pos = [], 2
    while pos < len(b):
        assert b[pos] == 0xFF
        m = b[pos + 1]
        ln = struct.unpack(">H", b[pos + 2:pos + 4])[0]
        out.append((pos, m, ln))
        if m == 0xDA:
            break
        pos += 2 + ln
    return out


base = open(os.path.join(D, "j_base_420.jpg"), "rb").read()
seg = markers(base)
# (a) fill bytes before markers, a comment segment, an APP segment with a short body
pos0 = seg[1][0]
edited = base[:pos0] + b"\xff\xff\xff" + b"\xff\xfe" + struct.pack(">H", 2 + 11) + b"hello world" + b"\xff\xe5" + struct.pack(">H", 4) + b"ab" + b"\xff" + base[pos0:]
open(os.path.join(D, "j_edit_fill_com.jpg"), "wb").write(edited)
# (b) entropy-coded data cut in the middle, EOI right after: the rest of the scan reads as zero bits
sos_off, _, sos_len = seg[-1]
data0 = sos_off + 2 + sos_len
cut = data0 + (len(base) - 2 - data0) // 2
while base[cut - 1] == 0xFF:
    cut += 1
open(os.path.join(D, "j_edit_cut_eoi.jpg"), "wb").write(base[:cut] + b"\xff\xd9")
# (c) zero bytes between the end of the scan and EOI
open(os.path.join(D, "j_edit_tail_zeros.jpg"), "wb").write(base[:-2] + b"\x00" * 7 + b"\xff\xd9")
# (d) the same stream declared as Adobe transform 2 on a 4-component image (YCCK)
cm = open(os.path.join(D, "j_cmyk.jpg"), "rb").read()
i = cm.index(b"Adobe")
assert cm[i + 11] == 0
open(os.path.join(D, "j_edit_ycck.jpg"), "wb").write(cm[:i + 11] + b"\x02" + cm[i + 12:])
# (e) 4 components without an Adobe segment: the first three are taken as YCbCr
a0 = cm.rindex(b"\xff\xee", 0, i)
ln = struct.unpack(">H", cm[a0 + 2:a0 + 4])[0]
open(os.path.join(D, "j_edit_4comp_noadobe.jpg"), "wb").write(cm[:a0] + cm[a0 + 2 + ln:])
# (f) 16-bit quantisation table: rewrite the first DQT of a file with precision 1
qoff, _, qlen = [x for x in seg if x[1] == 0xDB][0]
body = base[qoff + 4:qoff + 2 + qlen]
assert body[0] >> 4 == 0
tabs, k = b"", 0
while k < len(body):
    t = body[k] & 15
    tabs += bytes([0x10 | t]) + b"".join(struct.pack(">H", v) for v in body[k + 1:k + 65])
    k += 65
open(os.path.join(D, "j_edit_dqt16.jpg"), "wb").write(base[:qoff] + b"\xff\xdb" + struct.pack(">H", 2 + len(tabs)) + tabs + base[qoff + 2 + qlen:])
print("written to", D)
