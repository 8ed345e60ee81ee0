"""utils.video_utils (the frame source / sink behind video_test.py and the cv2 shim's VideoCapture / VideoWriter): a
Motion-JPEG AVI written here reads back frame for frame, the container is what other tools expect, and what cannot be
decoded says so.  No GPU."""
import io
import os
import struct

import numpy as np
import pytest


def _frames(n, size, seed=0):
    from PIL import Image
    rng = np.random.RandomState(seed)
    out = []
    for _ in range(n):
        small = rng.randint(0, 256, (size[1] // 10, size[0] // 10, 3)).astype(np.uint8)
        out.append(np.asarray(Image.fromarray(small).resize(size, Image.BICUBIC)))
    return out


def test_mjpeg_avi_round_trip_and_container_layout(tmp_path):
    from PIL import Image
    from yolov3_tensorflow_amd.utils import video_utils as V
    frames = _frames(6, (200, 150))
    path = str(tmp_path / 'clip.avi')
    with V.MjpegAviWriter(path, 29.97, (200, 150), quality=90) as w:
        for f in frames:
            w.write(f)
        with pytest.raises(ValueError, match='does not match'):
            w.write(frames[0][:100])
    r = V.open_video(path)
    assert (r.width, r.height, r.frame_count) == (200, 150, 6) and abs(r.fps - 29.97) < 1e-3
    for f in frames:
        got = r.read()
        buf = io.BytesIO()
        Image.fromarray(f).save(buf, format='JPEG', quality=90)        # exactly the JPEG the writer stored
        np.testing.assert_array_equal(got, np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert('RGB')))
    assert r.read() is None
    r.close()
    # the container: RIFF size, the hdrl / movi / idx1 triple, one keyframe index entry per frame pointing at a '00dc' chunk
    raw = open(path, 'rb').read()
    assert raw[:4] == b'RIFF' and raw[8:12] == b'AVI ' and struct.unpack('<I', raw[4:8])[0] == len(raw) - 8
    movi = raw.index(b'movi')
    idx = raw.index(b'idx1', movi)
    assert struct.unpack('<I', raw[idx + 4:idx + 8])[0] == 16 * 6
    for k in range(6):
        ckid, flags, off, size = struct.unpack('<4sIII', raw[idx + 8 + 16 * k:idx + 24 + 16 * k])
        assert ckid == b'00dc' and flags == 0x10
        assert raw[movi + off:movi + off + 4] == b'00dc' and struct.unpack('<I', raw[movi + off + 4:movi + off + 8])[0] == size
        assert raw[movi + off + 8:movi + off + 10] == b'\xff\xd8'          # a JPEG starts there
    assert struct.unpack('<I', raw[raw.index(b'avih') + 8 + 16:raw.index(b'avih') + 8 + 20])[0] == 6      # dwTotalFrames


def test_other_sources_and_refusals(tmp_path):
    from PIL import Image
    from yolov3_tensorflow_amd.utils import video_utils as V
    frames = _frames(3, (64, 48), seed=1)
    # a directory of frames, in sorted order
    os.makedirs(tmp_path / 'seq')
    for i, f in enumerate(frames):
        Image.fromarray(f).save(str(tmp_path / 'seq' / ('f%03d.png' % i)))
    r = V.open_video(str(tmp_path / 'seq'))
    assert (r.frame_count, r.width, r.height) == (3, 64, 48)
    for f in frames:
        np.testing.assert_array_equal(r.read(), f)
    # an animated image
    Image.fromarray(frames[0]).save(str(tmp_path / 'a.gif'), save_all=True, duration=40,
                                    append_images=[Image.fromarray(f) for f in frames[1:]])
    r = V.open_video(str(tmp_path / 'a.gif'))
    assert r.frame_count == 3 and abs(r.fps - 25.0) < 1e-6 and r.read().shape == (48, 64, 3)
    # an uncompressed AVI (bottom-up 24-bit DIB frames), built by hand
    w, h = 6, 4
    img = np.arange(h * w * 3, dtype=np.uint8).reshape(h, w, 3)
    stride = (w * 3 + 3) & ~3
    dib = b''.join(bytes(row[:, ::-1].tobytes()) + b'\0' * (stride - w * 3) for row in img[::-1])
    strh = b'vids' + b'DIB ' + struct.pack('<IHHIIIIIIII4H', 0, 0, 0, 0, 1, 10, 0, 1, len(dib), 0, 0, 0, 0, w, h)
    strf = struct.pack('<IiiHHIIiiII', 40, w, h, 1, 24, 0, len(dib), 0, 0, 0, 0)
    avih = struct.pack('<14I', 100000, 0, 0, 0, 1, 0, 1, len(dib), w, h, 0, 0, 0, 0)
    strl = b'strl' + b'strh' + struct.pack('<I', len(strh)) + strh + b'strf' + struct.pack('<I', len(strf)) + strf
    hdrl = b'hdrl' + b'avih' + struct.pack('<I', len(avih)) + avih + b'LIST' + struct.pack('<I', len(strl)) + strl
    movi = b'movi' + b'00db' + struct.pack('<I', len(dib)) + dib
    body = b'AVI ' + b'LIST' + struct.pack('<I', len(hdrl)) + hdrl + b'LIST' + struct.pack('<I', len(movi)) + movi
    open(str(tmp_path / 'raw.avi'), 'wb').write(b'RIFF' + struct.pack('<I', len(body)) + body)
    r = V.open_video(str(tmp_path / 'raw.avi'))
    assert (r.frame_count, r.fps) == (1, 10.0)
    np.testing.assert_array_equal(r.read(), img)
    # refusals name the reason
    open(str(tmp_path / 'x.mp4'), 'wb').write(b'\0\0\0\x18ftypmp42' + b'\0' * 64)
    with pytest.raises(V.VideoError, match='not decodable here'):
        V.open_video(str(tmp_path / 'x.mp4'))
    with pytest.raises(V.VideoError, match='no such file'):
        V.open_video(str(tmp_path / 'missing.avi'))
    bad = open(str(tmp_path / 'raw.avi'), 'rb').read().replace(struct.pack('<HHI', 1, 24, 0), struct.pack('<HH4s', 1, 24, b'H264'))
    open(str(tmp_path / 'h264.avi'), 'wb').write(bad)
    with pytest.raises(V.VideoError, match="codec b'H264'"):
        V.open_video(str(tmp_path / 'h264.avi'))


def test_cv2_shim_video_classes(tmp_path):
    """cv2.VideoCapture / VideoWriter / VideoWriter_fourcc as video_test.py:42-52,66,108-115 uses them (B,G,R frames)."""
    from yolov3_tensorflow_amd import compat
    compat.install()
    import cv2
    frames = _frames(3, (96, 64), seed=2)
    out = cv2.VideoWriter(str(tmp_path / 'o.avi'), cv2.VideoWriter_fourcc(*'MJPG'), 15, (96, 64))
    assert out.isOpened() and out.filename.endswith('o.avi')
    for f in frames:
        out.write(f[:, :, ::-1])
    out.release()
    cap = cv2.VideoCapture(str(tmp_path / 'o.avi'))
    assert cap.isOpened()
    assert [int(cap.get(k)) for k in (3, 4, 5, 7)] == [96, 64, 15, 3]
    ok, bgr = cap.read()
    assert ok and bgr.shape == (64, 96, 3) and np.abs(bgr[:, :, ::-1].astype(int) - frames[0].astype(int)).mean() < 6
    assert cap.read()[0] and cap.read()[0] and cap.read() == (False, None)
    cap.release()
    other = cv2.VideoWriter(str(tmp_path / 'r.mp4'), cv2.VideoWriter_fourcc('m', 'p', '4', 'v'), 15, (96, 64))
    assert other.filename.endswith('r.avi')
    other.release()
    closed = cv2.VideoCapture(str(tmp_path / 'nothing.avi'))
    assert not closed.isOpened() and closed.get(7) == 0 and closed.read() == (False, None)
