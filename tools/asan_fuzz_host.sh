#!/bin/bash
# Host-side sanitizer run (CPU only): the image decoders and the OBJ reader of pathtracer_amd/host under ASan + UBSan on
# mutated copies (truncations, byte noise, insertions, bit flips) of the test fixtures. usage: tools/asan_fuzz_host.sh [copies per file]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=$(mktemp -d)
cat > $W/drv.cpp <<'CPP'
#include "image_loader.h"
#include "obj_loader.h"
#include <cstdio>
#include <string>
using namespace prt_host;
int main(int argc, char** argv)
{
    int ok = 0;
    for(int i = 2; i < argc; ++i)
    {
        const std::string p = argv[i];
        if(p.size() > 4 && p.substr(p.size() - 4) == ".obj") { ObjData d; std::string err; ok += load_obj(p, argv[1], d, err); }
        else { Image img; ok += load_image(p, img); }
    }
    printf("%d of %d loaded\n", ok, argc - 2);
    return 0;
}
CPP
H=$ROOT/pathtracer_amd/host
g++ -std=c++17 -O1 -g -fwrapv -fsanitize=address,undefined -fno-sanitize-recover=all -I$H $W/drv.cpp $H/image_loader.cpp $H/jpeg_loader.cpp $H/obj_loader.cpp -o $W/drv
python3 - "$ROOT" "$W" "${1:-12}" <<'PY'
import glob, os, random, subprocess, sys
root, w, copies = sys.argv[1], sys.argv[2], int(sys.argv[3])
r = random.Random(1)
src = sorted(glob.glob(root + "/tests/golden/obj/textures/*")) + sorted(glob.glob(root + "/tests/golden/obj_cases/*.obj")) + [root + "/tests/golden/obj/mixed.obj"]
files = []
for s in src:
    data, ext = open(s, "rb").read(), os.path.splitext(s)[1]
    for k in range(copies):
        d, mode = bytearray(data), r.randint(0, 3)
        if mode == 0:
            d = d[:r.randint(0, len(d))]
        elif mode == 1:
            for _ in range(r.randint(1, 8)):
                d[r.randrange(len(d))] = r.randrange(256)
        elif mode == 2:
            i = r.randrange(len(d)); d[i:i] = bytes(r.randrange(256) for _ in range(r.randint(1, 64)))
        else:
            for _ in range(r.randint(1, 3)):
                i = r.randrange(len(d)); d[i] ^= 1 << r.randrange(8)
        fn = f"{w}/{os.path.basename(s)}.{k}{ext}"
        open(fn, "wb").write(bytes(d)); files.append(fn)
for i in range(0, len(files), 50):
    p = subprocess.run([w + "/drv", w + "/"] + files[i:i + 50], capture_output=True, text=True)
    if p.returncode != 0:
        print(p.stderr[-2000:]); sys.exit(1)
print(len(files), "mutated files: no sanitizer report")
PY
rm -rf $W
