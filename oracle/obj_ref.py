"""TEST INFRASTRUCTURE — a second, independent restatement of the reference's
Wavefront OBJ / MTL loader (/root/reference/src/obj.rs:12-235) in plain Python,
written from the Rust text with Python's own string operations so that it shares
no code with raytracer_amd/csrc/host/obj_load.cpp.  The tests load the meshes
the reference's checkout holds with both and require identical tables.  Never
imported by the product.

Returned triangles are in the loader's creation order (before any BVH build).
"""
from __future__ import annotations

import struct
from fractions import Fraction

import numpy as np

f32 = np.float32


def parse_f32(text: str) -> np.float32:
    """`str::parse::<f32>()`: the binary32 value nearest to the decimal string.
    Going through a double rounds twice; that can differ from rounding once only
    when the double lands exactly half-way between two binary32 values, which is
    settled with exact rational arithmetic."""
    d = float(text)
    bits = struct.unpack("<Q", struct.pack("<d", d))[0]
    if bits & ((1 << 29) - 1) == (1 << 28) and d == d and abs(d) != float("inf"):
        exact = Fraction(text)
        lo = f32(np.nextafter(f32(d), f32(-np.inf))) if Fraction(float(f32(d))) > exact else f32(d)
        hi = f32(np.nextafter(lo, f32(np.inf)))
        dl, dh = exact - Fraction(float(lo)), Fraction(float(hi)) - exact
        if dl != dh:
            return lo if dl < dh else hi
        return lo if (lo.view(np.uint32) & 1) == 0 else hi
    return f32(d)


def parse_isize(text: str):
    """`str::parse::<isize>().ok()`: an optional sign and decimal digits, else None."""
    body = text[1:] if text[:1] in ("+", "-") else text
    if not body or not body.isascii() or not body.isdigit():
        return None
    return int(text)


def vec3(text: str):
    """vec3.rs:541-547: three numbers separated by single blanks."""
    it = text.split(" ")
    return [parse_f32(it[0]), parse_f32(it[1]), parse_f32(it[2])]


def rust_lines(text: str):
    """`str::lines`: split at \\n, drop one trailing \\r per line, no empty last line."""
    parts = text.split("\n")
    if parts and parts[-1] == "":
        parts.pop()
    return [p[:-1] if p.endswith("\r") else p for p in parts]


class Materials:
    """The interner of config.rs:157-178: first-seen order, equality lookup."""

    def __init__(self):
        self.items = []

    def intern(self, item) -> int:
        if item in self.items:
            return self.items.index(item)
        self.items.append(item)
        return len(self.items) - 1


GREY = ("lambertian", "solid", (f32(0.5), f32(0.5), f32(0.5)))


def parse_materials(materials: Materials, mtl_text, parent: str, read_texture):
    """obj.rs:196-235."""
    name_index = {}
    if mtl_text is None:
        return name_index
    for section in mtl_text.split("newmtl ")[1:]:
        ls = rust_lines(section)
        name, rest = ls[0], ls[1:]
        colour = next((tuple(vec3(l[3:])) for l in rest if l.startswith("Kd")), None)
        texture = next((l[7:] for l in rest if l.startswith("map_Kd")), None)
        if texture is not None:
            item = ("lambertian", "texture", read_texture(f"{parent}/{texture}"))
        elif colour is not None:
            item = ("lambertian", "solid", colour)
        else:
            item = GREY
        name_index[name] = materials.intern(item)
    return name_index


def dot(a, b):
    return f32(f32(f32(a[0] * b[0]) + f32(a[1] * b[1])) + f32(a[2] * b[2]))


def normalize(v):
    n = np.sqrt(dot(v, v))
    return [f32(v[0] / n), f32(v[1] / n), f32(v[2] / n)]


def parse(path: str, materials: Materials, read_text, read_texture):
    """obj.rs:12-192.  `read_text(path)` returns a file's text, `read_texture(path)`
    anything hashable that stands for a loaded P6 file.  Returns a dict of lists:
    triangles (a, b, c corner positions), material, uv (or None), normals (or None),
    bary (or None)."""
    text = read_text(path)
    ls = rust_lines(text)
    parent = path[: path.rfind("/") + 1]
    mtl_line = next((l for l in ls if l.startswith("mtllib")), None)
    mtl_text = read_text(f"{parent}/{mtl_line[7:]}") if mtl_line is not None else None
    name_index = parse_materials(materials, mtl_text, parent, read_texture)

    vertices = [vec3(l[2:].strip()) for l in ls if l.startswith("v ")]
    uvs = []
    for l in ls:
        if l.startswith("vt"):
            it = l[3:].strip().split(" ")
            uvs.append([parse_f32(it[0]), parse_f32(it[1])])
    normals = [vec3(l[3:].strip()) for l in ls if l.startswith("vn")]

    out = {"corners": [], "material": [], "uv": [], "normals": [], "bary": []}
    for piece in text.split("usemtl "):
        pl = rust_lines(piece)
        material = name_index.get(pl[0]) if pl else None
        if material is None:
            material = materials.intern(GREY)
        for l in pl[1:]:
            if not l.startswith("f"):
                continue
            corners = []
            for part in l[1:].split():
                idx = []
                for s, n in zip(part.split("/"), (len(vertices), len(uvs), len(normals))):
                    v = parse_isize(s)
                    idx.append(None if v is None else (n + v if v < 0 else v - 1))
                idx += [None] * (3 - len(idx))
                corners.append((vertices[idx[0]], None if idx[1] is None else uvs[idx[1]],
                                None if idx[2] is None else normals[idx[2]]))
            (v1, t1, n1) = corners[0]
            for (v2, t2, n2), (v3, t3, n3) in zip(corners[1:], corners[2:]):  # fan, obj.rs:116-117
                uv = [t1, t2, t3] if t1 is not None and t2 is not None and t3 is not None else None
                nn = [normalize(n) for n in (n1, n2, n3)] if n1 is not None and n2 is not None and n3 is not None else None
                bary = None
                if uv is not None or nn is not None:  # obj.rs:142-153
                    e1 = [f32(v2[k] - v1[k]) for k in range(3)]
                    e2 = [f32(v3[k] - v1[k]) for k in range(3)]
                    d00, d01, d11 = dot(e1, e1), dot(e1, e2), dot(e2, e2)
                    bary = [d00, d01, d11, f32(f32(d00 * d11) - f32(d01 * d01))]
                out["corners"].append((v1, v2, v3))
                out["material"].append(material)
                out["uv"].append(uv)
                out["normals"].append(nn)
                out["bary"].append(bary)
    return out
