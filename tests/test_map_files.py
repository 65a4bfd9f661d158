"""The map files either side of the render producers (include/nmi_host.h, host/nmi_map.cpp): OBJ meshes, XYZ clouds with their
offset file, BMP textures, read with the grammar of the reference's loaders (objloader.cpp:140-264, texture.cpp:31-86).
The reference tree holds no such files, so the cases are written here.  Host only."""
import struct

import numpy as np
import pytest

from orbslam2_nmi_amd import hostapi as H


def test_obj_expands_faces_to_per_corner_arrays(tmp_path):
    p = tmp_path / "m.obj"
    p.write_text("# a comment line that is skipped\n"
                 "mtllib whatever.mtl\n"
                 "v 0 0 0\nv 1 0 0\nv 1 1 0.5\nv 0 1 -2.25\n"
                 "vt 0 0\nvt 1 0\nvt 1 1\nvt 0.25 0.75\n"
                 "vn 0 0 1\n"
                 "usemtl m\ns off\n"
                 "f 1/1 2/2 3/3\n"
                 "f 1/4 3/3 4/1\n")
    xyz, uv = H.load_obj(p)
    assert xyz.shape == (6, 3) and uv.shape == (6, 2) and xyz.dtype == np.float32
    v = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0.5], [0, 1, -2.25]], np.float32)
    t = np.array([[0, 0], [1, 0], [1, 1], [0.25, 0.75]], np.float32)
    assert np.array_equal(xyz, v[[0, 1, 2, 0, 2, 3]])
    assert np.array_equal(uv, t[[0, 1, 2, 3, 2, 0]])


def test_obj_faces_may_come_before_the_lists_they_index(tmp_path):
    """Indices are resolved after the whole file has been read (objloader.cpp:199-214)."""
    p = tmp_path / "m.obj"
    p.write_text("f 1/1 2/1 3/1\nv 1 2 3\nv 4 5 6\nv 7 8 9\nvt 0.5 0.5\n")
    xyz, uv = H.load_obj(p)
    assert np.array_equal(xyz, np.array([[1, 2, 3], [4, 5, 6], [7, 8, 9]], np.float32))
    assert np.array_equal(uv, np.full((3, 2), 0.5, np.float32))


@pytest.mark.parametrize("face, code", [("f 1/1/1 2/2/1 3/3/1", -2),   # normals in the faces: the reference's "can't be read by this simple parser"
                                        ("f 1 2 3", -2),
                                        ("f 1/1 2/2 3/3 4/4", None),     # a quad: the first three pairs are a face; "4/4" becomes a skipped line
                                        ("f 1/1 2/2 9/3", -3), ("f 1/1 2/2 3/0", -3)])
def test_obj_errors(tmp_path, face, code):
    p = tmp_path / "m.obj"
    p.write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\n" + face + "\n")
    if code is None:
        xyz, uv = H.load_obj(p)
        assert xyz.shape == (3, 3)
    else:
        with pytest.raises(ValueError, match=str(code)):
            H.load_obj(p)
    with pytest.raises(ValueError, match="-5"):
        H.load_obj(tmp_path / "missing.obj")


def test_empty_obj_is_an_empty_mesh(tmp_path):
    p = tmp_path / "m.obj"
    p.write_text("# nothing\n")
    xyz, uv = H.load_obj(p)
    assert xyz.shape == (0, 3) and uv.shape == (0, 2)


def test_xyz_subtracts_the_offset_in_double_and_scales_colours(tmp_path):
    cloud, off = tmp_path / "c.xyz", tmp_path / "c.offset"
    off.write_text("2683000.25 1250000.5\n400\n")
    pts = np.array([[2683001.375, 1250002.75, 401.5, 255, 128, 0],
                    [2683000.25, 1250000.5, 400.0, 64, 32, 16],
                    [2682990.0, 1249999.0, 399.125, 1, 2, 3]])
    cloud.write_text("\n".join(" ".join(repr(float(v)) for v in row) for row in pts) + "\n")   # ends in white space, like the reference's own files may
    xyz, red, rgb = H.load_xyz(cloud, off)
    want = (pts[:, :3] - [2683000.25, 1250000.5, 400.0]).astype(np.float32)   # differences exact in double; a float32 of the raw coordinate is not
    assert xyz.shape == (3, 3) and np.array_equal(xyz, want)
    assert np.array_equal(rgb, (pts[:, 3:].astype(np.float32) * np.float32(1 / 256)))
    assert np.array_equal(red, rgb[:, 0])
    assert not np.array_equal(xyz, (pts[:, :3].astype(np.float32) - np.array([2683000.25, 1250000.5, 400.0], np.float32)))


def test_xyz_errors(tmp_path):
    cloud, off = tmp_path / "c.xyz", tmp_path / "c.offset"
    off.write_text("0 0 0")
    cloud.write_text("1 2 3 4 5 6\n7 8 9 10\n")
    with pytest.raises(ValueError, match="-2"):
        H.load_xyz(cloud, off)
    cloud.write_text("")
    xyz, red, rgb = H.load_xyz(cloud, off)
    assert xyz.shape == (0, 3) and red.shape == (0,)
    off.write_text("1 2")
    with pytest.raises(ValueError, match="-2"):
        H.load_xyz(cloud, off)
    with pytest.raises(ValueError, match="-5"):
        H.load_xyz(cloud, tmp_path / "missing.offset")


def bmp_bytes(rgb, size_field=None, data_offset=54, bpp=24, compression=0, magic=b"BM"):
    h, w = rgb.shape[:2]
    data = rgb.tobytes()
    head = bytearray(54)
    head[0:2] = magic
    struct.pack_into("<I", head, 0x02, 54 + len(data))
    struct.pack_into("<I", head, 0x0A, data_offset)
    struct.pack_into("<I", head, 0x0E, 40)
    struct.pack_into("<ii", head, 0x12, w, h)
    struct.pack_into("<HH", head, 0x1A, 1, bpp)
    struct.pack_into("<I", head, 0x1E, compression)
    struct.pack_into("<I", head, 0x22, len(data) if size_field is None else size_field)
    return bytes(head) + data


def test_bmp_bytes_come_back_in_file_order(tmp_path):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (6, 8, 3), dtype=np.uint8)
    p = tmp_path / "t.bmp"
    p.write_bytes(bmp_bytes(img))
    assert np.array_equal(H.load_bmp(p), img)
    p.write_bytes(bmp_bytes(img, size_field=0, data_offset=0))   # "some BMP files are misformatted, guess missing information" (texture.cpp:65-67)
    assert np.array_equal(H.load_bmp(p), img)
    # the reference reads the image from where the header ended, whatever offset the header names (texture.cpp:73)
    p.write_bytes(bmp_bytes(img, data_offset=1078))
    assert np.array_equal(H.load_bmp(p), img)


def test_bmp_feeds_the_texture_builder(tmp_path):
    """The loader's output is the argument of nmi_texture_create; its luma is 0.299 byte0 + 0.587 byte1 + 0.114 byte2 (oracle: mip_luma)."""
    from oracle import mesh_oracle_np as mo
    img = np.random.default_rng(4).integers(0, 256, (4, 4, 3), dtype=np.uint8)
    p = tmp_path / "t.bmp"
    p.write_bytes(bmp_bytes(img))
    levels = mo.mip_luma(H.load_bmp(p))
    assert len(levels) == 3 and levels[0].shape == (4, 4)
    c = img.astype(np.float32) / np.float32(255)
    assert np.array_equal(levels[0], (np.float32(0.299) * c[..., 0] + np.float32(0.587) * c[..., 1]) + np.float32(0.114) * c[..., 2])


@pytest.mark.parametrize("kw", [dict(bpp=32), dict(compression=1), dict(magic=b"PN")])
def test_bmp_rejects_what_the_reference_rejects(tmp_path, kw):
    p = tmp_path / "t.bmp"
    p.write_bytes(bmp_bytes(np.zeros((2, 2, 3), np.uint8), **kw))
    with pytest.raises(ValueError, match="-2"):
        H.load_bmp(p)


def test_bmp_truncated(tmp_path):
    p = tmp_path / "t.bmp"
    p.write_bytes(bmp_bytes(np.zeros((4, 4, 3), np.uint8))[:-5])
    with pytest.raises(ValueError, match="-2"):
        H.load_bmp(p)
    p.write_bytes(b"BM" + bytes(20))
    with pytest.raises(ValueError, match="-2"):
        H.load_bmp(p)
    with pytest.raises(ValueError, match="-5"):
        H.load_bmp(tmp_path / "missing.bmp")
