"""A minimal grey PNG writer for the scripts that build synthetic workdirs (zlib level 1: the files only have to be valid)."""
import struct, zlib


def write_png(path, img):
    hh, ww = img.shape
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(hh))

    def chunk(t, d):
        c = struct.pack(">I", len(d)) + t + d
        return c + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", ww, hh, 8, 0, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 1)) + chunk(b"IEND", b""))
