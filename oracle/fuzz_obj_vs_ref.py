#!/usr/bin/env python3
"""TEST INFRASTRUCTURE, build container only (needs oracle/_ref/ref_host_dump, i.e. /root/reference).
Random OBJ / MTL text through the loader the reference uses (tinyobj::LoadObj via ref_host_dump --obj) and through
pathtracer_amd/host/obj_loader.cpp (prth_obj_dump_json): every float by bits, every index, name and message.

    python oracle/fuzz_obj_vs_ref.py [cases] [seed]
"""
import json
import os
import random
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from pathtracer_amd import hostapi  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "ref_host_dump")


def number(r):
    k = r.random()
    if k < 0.45:
        return f"{r.uniform(-5, 5):.{r.randint(0, 12)}f}"
    if k < 0.6:
        return str(r.randint(-9, 9))
    if k < 0.8:
        return f"{r.uniform(-5, 5):.{r.randint(0, 9)}e}".replace("e", r.choice("eE"))
    return r.choice([".5", "-.5", "5.", "+2", "-0", "1e", "1e+", "1e-2", "2E3", "1.5x", "abc", "1e0", "0.1234567891234", "-0.0",
                     "3.e2", "1.25e+1", "00012", "1..2", "1e1.5", "--1", "+-1", "1e-0", "123456789012345678", "1e30", "1e-30"])


def polygon(r, nv):
    n = r.randint(1, 9)
    plane = r.random()
    if plane < 0.7:  # corners from a small pool of vertices: repeats, collinear runs
        return [r.randint(1, nv) for _ in range(n)]
    return [r.randint(-nv, -1) for _ in range(n)]


def corner(r, i, nvt, nvn):
    k = r.random()
    if k < 0.5 or (nvt == 0 and nvn == 0):
        return str(i)
    if k < 0.65 and nvt:
        return f"{i}/{r.randint(1, nvt)}"
    if k < 0.8 and nvn:
        return f"{i}//{r.randint(1, nvn)}"
    if nvt and nvn:
        return f"{i}/{r.randint(-nvt, nvt) or 1}/{r.randint(1, nvn)}"
    return str(i)


def make_case(r):
    mtl = []
    names = ["m%d" % k for k in range(r.randint(0, 3))]
    if r.random() < 0.3:
        mtl.append(f"Kd {number(r)} {number(r)} {number(r)}")
    for n in names:
        mtl.append(f"newmtl {n}" + r.choice(["", " ", "  "]))
        for _ in range(r.randint(0, 5)):
            key = r.choice(["Ka", "Kd", "Ks", "Ke", "Kt", "Tf", "Ni", "Ns", "d", "Tr", "Pr", "Pm", "illum", "map_Kd", "map_Ks",
                            "map_Bump", "bump", "Ps", "zzz"])
            if key.startswith("map") or key == "bump":
                opts = r.choice(["", "-bm 0.5 ", "-o 1 2 3 ", "-s 1 1 ", "-clamp on ", "-mm 0 1 ", "-bm", "-type cube "])
                mtl.append(f"{key} {opts}" + r.choice(["a.png", "b c.png", "dir\\\\x.png", ""]))
            elif key in ("Ka", "Kd", "Ks", "Ke", "Kt", "Tf"):
                mtl.append(f"{key} " + " ".join(number(r) for _ in range(r.randint(1, 4))))
            else:
                mtl.append(f"{key}{r.choice([' ', '  ', chr(9)])}{number(r)}")
    obj = []
    if r.random() < 0.8:
        obj.append("mtllib " + r.choice(["", "nofile.mtl "]) + "case.mtl")
    nv = nvt = nvn = 0
    nl = r.choice(["\n", "\n", "\r\n", "\r"])
    for _ in range(r.randint(3, 14)):
        k = r.random()
        if k < 0.45 or nv < 3:
            grid = r.random() < 0.5  # points on a coarse grid: exact collinearity / coincidence happens
            xyz = [str(r.randint(0, 3)) if grid else number(r) for _ in range(3)]
            if r.random() < 0.3:
                xyz[r.randint(0, 2)] = "0"
            col = [number(r) for _ in range(r.choice([0, 0, 0, 1, 3]))]
            obj.append("v " + " ".join(xyz + col))
            nv += 1
        elif k < 0.5:
            obj.append("vt " + " ".join(number(r) for _ in range(r.randint(1, 3))))
            nvt += 1
        elif k < 0.55:
            obj.append("vn " + " ".join(number(r) for _ in range(3)))
            nvn += 1
        elif k < 0.85:
            obj.append(r.choice(["f ", "f  ", " f "]) + r.choice([" ", "  ", "\t"]).join(corner(r, i, nvt, nvn) for i in polygon(r, nv)))
        elif k < 0.9:
            obj.append("usemtl " + r.choice(names + ["nosuch", "m0 "]))
        elif k < 0.94:
            obj.append(r.choice(["g", "g ", "g a", "g a b", "o", "o x y ", "o  z"]))
        elif k < 0.97:
            obj.append(r.choice(["l 1 2", "p 1", "s 1", "s off", "# c", "", "  ", "vp 1 2"]))
        else:
            obj.append("f " + " ".join(str(r.randint(0, nv + 1)) for _ in range(r.randint(3, 5))))  # 0 / out of range
    return nl.join(obj) + r.choice(["", nl]), "\n".join(mtl) + "\n"


def f32(x):
    return np.asarray(x, np.float32)


def same(a, b):
    if a["ok"] != b["ok"] or a["warn"] != b["warn"] or a["err"] != b["err"]:
        return "ok/warn/err"
    for k in ("vertices", "normals", "texcoords", "colors"):
        if f32(a[k]).tobytes() != f32(b[k]).tobytes():
            return k
    if [(s["name"], s["indices"], s["material_ids"]) for s in a["shapes"]] != [(s["name"], s["indices"], s["material_ids"]) for s in b["shapes"]]:
        return "shapes"
    if len(a["materials"]) != len(b["materials"]):
        return "material count"
    for x, y in zip(a["materials"], b["materials"]):
        for k, v in y.items():
            if (x[k] != v) if isinstance(v, str) else (f32(x[k]).tobytes() != f32(v).tobytes()):
                return f"material {y['name']} {k}"
    return None


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    r = random.Random(seed)
    L = hostapi.lib()
    bad = 0
    with tempfile.TemporaryDirectory() as d:
        for k in range(cases):
            obj, mtl = make_case(r)
            open(os.path.join(d, "case.obj"), "w", newline="").write(obj)
            open(os.path.join(d, "case.mtl"), "w", newline="").write(mtl)
            subprocess.check_call([REF, "--obj", os.path.join(d, "case.obj"), d + "/", os.path.join(d, "ref.json")])
            L.prth_obj_dump_json(os.path.join(d, "case.obj").encode(), (d + "/").encode(), os.path.join(d, "mine.json").encode())
            ref = json.load(open(os.path.join(d, "ref.json")), parse_int=float)["obj"]
            mine = json.load(open(os.path.join(d, "mine.json")), parse_int=float)
            why = same(mine, ref)
            if why:
                bad += 1
                keep = os.path.join(tempfile.gettempdir(), f"objfuzz_{seed}_{k}")
                os.makedirs(keep, exist_ok=True)
                for n in ("case.obj", "case.mtl", "ref.json", "mine.json"):
                    open(os.path.join(keep, n), "wb").write(open(os.path.join(d, n), "rb").read())
                print(f"case {k}: differs in {why} -> {keep}")
                if bad >= 5:
                    break
    print(f"{cases} cases, seed {seed}: {bad} differences")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
