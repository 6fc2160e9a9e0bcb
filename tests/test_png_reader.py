"""host/png_gray8.h -- the PNG reader of the file-based odometry entry (8-bit grayscale, non-interlaced): compiled into a tiny
program here and run on files written by a from-scratch encoder that forces every filter type on every row position (first row
included), splits the stream over several IDAT chunks and adds ancillary chunks; the bytes must be the image.  Error cases: a
16-bit image, a truncated file, a buffer that is too small."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = r'''
#include "png_gray8.h"
#include <cstdio>
int main(int argc, char **argv) {
  try {
    int w = 0, h = 0;
    if (argc > 2) {  // a buffer one byte short
      rsxhost::PngScratch s;
      rsxhost::read_png_gray8_into(argv[1], nullptr, 0, &w, &h, s);
      std::vector<uint8_t> small((size_t)w * h - 1);
      rsxhost::read_png_gray8_into(argv[1], small.data(), small.size(), &w, &h, s);
      return 3;
    }
    const std::vector<uint8_t> img = rsxhost::read_png_gray8(argv[1], &w, &h);
    std::printf("%d %d\n", w, h);
    std::fwrite(img.data(), 1, img.size(), stdout);
    return 0;
  } catch (const std::exception &e) {
    std::fprintf(stderr, "%s\n", e.what());
    return 2;
  }
}
'''


@pytest.fixture(scope="module")
def reader(tmp_path_factory):
    d = tmp_path_factory.mktemp("png")
    (d / "main.cpp").write_text(SRC)
    exe = d / "reader"
    subprocess.run(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "navtech-radar-slam_amd", "host"), str(d / "main.cpp"), "-lz", "-o", str(exe)],
                   check=True)
    return str(exe)


def _chunk(kind, data):
    return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data) & 0xffffffff)


def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)


def write_png(path, img, filters, n_idat=1, bit_depth=8):
    h, w = img.shape
    raw = bytearray()
    for y in range(h):
        f = filters[y]
        row = img[y].astype(np.int32)
        up = img[y - 1].astype(np.int32) if y else np.zeros(w, np.int32)
        left = np.concatenate([[0], row[:-1]])
        ul = np.concatenate([[0], up[:-1]])
        if f == 0:
            pred = np.zeros(w, np.int32)
        elif f == 1:
            pred = left
        elif f == 2:
            pred = up
        elif f == 3:
            pred = (left + up) >> 1
        else:
            pred = np.array([_paeth(int(a), int(b), int(c)) for a, b, c in zip(left, up, ul)], dtype=np.int32)
        raw.append(f)
        raw += bytes(((row - pred) & 0xff).astype(np.uint8))
    z = zlib.compress(bytes(raw), 6)
    cuts = np.linspace(0, len(z), n_idat + 1).astype(int)
    out = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, bit_depth, 0, 0, 0, 0)) + _chunk(b"tEXt", b"Comment\0radar")
    for a, b in zip(cuts[:-1], cuts[1:]):
        out += _chunk(b"IDAT", z[a:b])
    out += _chunk(b"IEND", b"")
    with open(path, "wb") as fh:
        fh.write(out)


def _read(reader, path):
    r = subprocess.run([reader, str(path)], capture_output=True)
    assert r.returncode == 0, r.stderr
    head, _, body = r.stdout.partition(b"\n")
    w, h = (int(x) for x in head.split())
    return np.frombuffer(body, dtype=np.uint8).reshape(h, w)


@pytest.mark.parametrize("n_idat", [1, 2, 7])
def test_every_filter_on_every_row_position(reader, tmp_path, n_idat):
    rng = np.random.default_rng(3)
    for w, h in ((1, 1), (2, 7), (97, 23), (3371, 12)):
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
        img[h // 2] = 255                          # saturated and zero rows: the wrap of the byte arithmetic
        if h > 2:
            img[h // 2 - 1] = 0
        for start in range(5):
            filters = [(start + y) % 5 for y in range(h)]          # every filter type is the first row's once
            p = tmp_path / f"f{w}_{h}_{start}_{n_idat}.png"
            write_png(p, img, filters, n_idat=n_idat)
            assert np.array_equal(_read(reader, p), img), (w, h, start)
        smooth = (np.add.outer(np.arange(h), np.arange(w)) % 256).astype(np.uint8)   # what the filters are made for
        p = tmp_path / f"s{w}_{h}_{n_idat}.png"
        write_png(p, smooth, [4] * h, n_idat=n_idat)
        assert np.array_equal(_read(reader, p), smooth)


def test_a_file_written_by_pillow(reader, tmp_path):
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(4)
    img = (rng.gamma(2.0, 20.0, (40, 3371)).clip(0, 255)).astype(np.uint8)
    p = tmp_path / "pil.png"
    Image.fromarray(img, mode="L").save(str(p))
    assert np.array_equal(_read(reader, p), img)


def test_errors(reader, tmp_path):
    img = np.zeros((4, 4), dtype=np.uint8)
    p16 = tmp_path / "deep.png"
    write_png(p16, img, [0] * 4, bit_depth=16)
    r = subprocess.run([reader, str(p16)], capture_output=True)
    assert r.returncode == 2 and b"8-bit grayscale" in r.stderr
    good = tmp_path / "good.png"
    write_png(good, img, [1] * 4)
    cut = tmp_path / "cut.png"
    cut.write_bytes(good.read_bytes()[:-20])
    r = subprocess.run([reader, str(cut)], capture_output=True)
    assert r.returncode == 2
    r = subprocess.run([reader, str(good), "small"], capture_output=True)
    assert r.returncode == 2 and b"larger than the buffer" in r.stderr
    r = subprocess.run([reader, str(tmp_path / "absent.png")], capture_output=True)
    assert r.returncode == 2 and b"cannot open" in r.stderr
