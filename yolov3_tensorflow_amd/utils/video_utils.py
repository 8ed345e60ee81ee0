# coding: utf-8
"""Frame sources and sinks for the video demo (reference video_test.py:42-52,66,108-115: cv2.VideoCapture /
cv2.VideoWriter) without OpenCV or ffmpeg: what this stack can decode and encode itself is JPEG (Pillow), so the video
format is Motion-JPEG in an AVI (RIFF) container - every frame an independent JPEG, playable by the usual players and
written by `ffmpeg -c:v mjpeg` or OpenCV's own VideoWriter_fourcc(*'MJPG').

    open_video(path)   -> a reader with .width .height .fps .frame_count and .read() -> RGB uint8 HxWx3 or None
                          (an MJPEG / uncompressed AVI, any multi-frame image Pillow opens - GIF, APNG, TIFF, WebP -,
                           or a directory of image files taken in sorted order)
    MjpegAviWriter(path, fps, (width, height)).write(rgb) ... .close()

H.264 / MPEG-4 streams are NOT decoded here (that is a codec library's job); such a file raises with that message.
"""
from __future__ import division, print_function

import io
import os
import struct

import numpy as np

_IMAGE_SUFFIXES = ('.jpg', '.jpeg', '.png', '.bmp', '.webp', '.tif', '.tiff')


class VideoError(IOError):
    pass


def _chunks(f, start, end):
    """(fourcc, payload offset, payload size) of the RIFF chunks laid out in [start, end)."""
    pos = start
    while pos + 8 <= end:
        f.seek(pos)
        head = f.read(8)
        if len(head) < 8:
            return
        fourcc, size = head[:4], struct.unpack('<I', head[4:])[0]
        yield fourcc, pos + 8, size
        pos += 8 + size + (size & 1)              # chunks are padded to even sizes


class AviReader(object):
    """Sequential reader of the first video stream of an AVI file whose frames are JPEGs (MJPG / mjpg / jpeg handlers) or
    uncompressed 24-bit DIBs.  The frame table comes from walking the 'movi' list, so files without an index work too."""

    def __init__(self, path):
        self.path = path
        self._f = open(path, 'rb')
        try:
            self._open()
        except Exception:
            self._f.close()
            raise

    def _open(self):
        path = self.path
        head = self._f.read(12)
        if len(head) < 12 or head[:4] != b'RIFF' or head[8:12] != b'AVI ':
            raise VideoError("%s is not an AVI (RIFF) file" % path)
        size = os.path.getsize(path)
        self.width = self.height = 0
        self.fps = 0.0
        self._codec = None
        self._frames = []           # (offset, size) of the stream-0 video chunks
        self._walk(12, size, depth=0)
        if self._codec is None:
            raise VideoError("%s: no video stream header found" % path)
        name = self._codec.strip(b'\0 ').upper()
        if name in (b'MJPG', b'JPEG', b'AVRN', b'LJPG', b'JPGL'):
            self._decode = self._decode_jpeg
        elif name in (b'', b'DIB', b'RGB', b'RAW'):
            self._decode = self._decode_dib
        else:
            raise VideoError("%s: video codec %r is not decodable here (Motion-JPEG and uncompressed AVI are); transcode with "
                             "`ffmpeg -i in -c:v mjpeg -q:v 3 out.avi`" % (path, self._codec))
        self.frame_count = len(self._frames)
        self._next = 0

    def _walk(self, start, end, depth):
        stream_kind = None
        for fourcc, off, size in _chunks(self._f, start, end):
            if fourcc == b'LIST':
                self._f.seek(off)
                kind = self._f.read(4)
                if kind == b'movi':
                    self._movi(off + 4, off + size)
                elif depth < 4:
                    self._walk(off + 4, off + size, depth + 1)
            elif fourcc == b'avih' and size >= 40:
                self._f.seek(off)
                v = struct.unpack('<10I', self._f.read(40))
                if v[0]:
                    self.fps = 1e6 / v[0]
                self.width, self.height = v[8], v[9]
            elif fourcc == b'strh' and size >= 32:
                self._f.seek(off)
                raw = self._f.read(32)
                stream_kind = raw[:4]
                if stream_kind == b'vids' and self._codec is None:
                    scale, rate = struct.unpack('<II', raw[20:28])
                    if scale and rate:
                        self.fps = rate / float(scale)
            elif fourcc == b'strf' and stream_kind == b'vids' and self._codec is None and size >= 40:
                self._f.seek(off)
                bih = struct.unpack('<IiiHHI', self._f.read(20))
                self.width, self.height, self._bits = bih[1], abs(bih[2]), bih[4]
                self._bottom_up = bih[2] > 0
                self._codec = struct.pack('<I', bih[5]) if bih[5] else b''

    def _movi(self, start, end):
        for fourcc, off, size in _chunks(self._f, start, end):
            if fourcc == b'LIST':                                   # 'rec ' groups
                self._movi(off + 4, off + size)
            elif fourcc[:2] == b'00' and fourcc[2:] in (b'dc', b'db') and size:
                self._frames.append((off, size))

    def _decode_jpeg(self, data):
        from PIL import Image
        return np.asarray(Image.open(io.BytesIO(data)).convert('RGB'))

    def _decode_dib(self, data):
        if self._bits != 24:
            raise VideoError("%s: uncompressed frames of %d bits per pixel are not supported (24 is)" % (self.path, self._bits))
        stride = (self.width * 3 + 3) & ~3
        rows = np.frombuffer(data, np.uint8, stride * self.height).reshape(self.height, stride)[:, :self.width * 3]
        img = rows.reshape(self.height, self.width, 3)[:, :, ::-1]          # B,G,R -> R,G,B
        return np.ascontiguousarray(img[::-1] if self._bottom_up else img)

    def read(self):
        if self._next >= len(self._frames):
            return None
        off, size = self._frames[self._next]
        self._next += 1
        self._f.seek(off)
        return self._decode(self._f.read(size))

    def close(self):
        self._f.close()


class ImageSequenceReader(object):
    """A multi-frame image file (GIF, APNG, multi-page TIFF, animated WebP) or a directory of images as a video."""

    def __init__(self, path, fps=25.0):
        from PIL import Image
        self.path, self.fps = path, float(fps)
        if os.path.isdir(path):
            self._files = sorted(os.path.join(path, n) for n in os.listdir(path) if n.lower().endswith(_IMAGE_SUFFIXES))
            if not self._files:
                raise VideoError("%s holds no image files" % path)
            self._im = None
            self.frame_count = len(self._files)
            with Image.open(self._files[0]) as first:
                self.width, self.height = first.size
        else:
            self._files = None
            self._im = Image.open(path)
            self.frame_count = int(getattr(self._im, 'n_frames', 1))
            self.width, self.height = self._im.size
            duration = self._im.info.get('duration')
            if duration:
                self.fps = 1000.0 / float(duration)
        self._next = 0

    def read(self):
        from PIL import Image
        if self._next >= self.frame_count:
            return None
        if self._files is not None:
            with Image.open(self._files[self._next]) as im:
                frame = np.asarray(im.convert('RGB'))
        else:
            self._im.seek(self._next)
            frame = np.asarray(self._im.convert('RGB'))
        self._next += 1
        return frame

    def close(self):
        if self._im is not None:
            self._im.close()


def open_video(path):
    """The reader for `path` (see the module docstring); raises VideoError for what cannot be decoded here."""
    if os.path.isdir(path):
        return ImageSequenceReader(path)
    if not os.path.exists(path):
        raise VideoError("no such file: %s" % path)
    with open(path, 'rb') as f:
        head = f.read(12)
    if head[:4] == b'RIFF' and head[8:12] == b'AVI ':
        return AviReader(path)
    if head[4:8] == b'ftyp' or head[:4] == b'\x1aE\xdf\xa3':
        raise VideoError("%s is an MP4 / Matroska file: its codecs (H.264, MPEG-4, VP9 ...) are not decodable here; transcode "
                         "with `ffmpeg -i in -c:v mjpeg -q:v 3 out.avi`" % path)
    try:
        return ImageSequenceReader(path)
    except Exception as e:       # noqa: BLE001 - Pillow raises several types for unknown data
        raise VideoError("%s: not an AVI file and not an image Pillow can open (%s)" % (path, e))


class MjpegAviWriter(object):
    """Motion-JPEG AVI writer: header with placeholders, one '00dc' chunk per frame, 'idx1' index and the patched header
    on close().  Frames are RGB uint8 HxWx3 of exactly `size` = (width, height)."""

    def __init__(self, path, fps, size, quality=90):
        self.path, self.fps, self.quality = path, float(fps) if fps else 25.0, int(quality)
        self.width, self.height = int(size[0]), int(size[1])
        self._f = open(path, 'wb')
        self._index = []            # (offset relative to the 'movi' fourcc, size)
        self._largest = 0
        self._write_header(0)
        self._movi_fourcc_at = self._f.tell() - 4
        self.frame_count = 0

    def _write_header(self, frames):
        rate, scale = int(round(self.fps * 1000)), 1000
        avih = struct.pack('<14I', int(round(1e6 / self.fps)), 0, 0, 0x10, frames, 0, 1, self._largest, self.width,
                           self.height, 0, 0, 0, 0)
        strh = b'vids' + b'MJPG' + struct.pack('<IHHIIIIIIII4H', 0, 0, 0, 0, scale, rate, 0, frames, self._largest,
                                                0xFFFFFFFF, 0, 0, 0, self.width, self.height)
        strf = struct.pack('<IiiHH4sIiiII', 40, self.width, self.height, 1, 24, b'MJPG', self.width * self.height * 3, 0, 0,
                           0, 0)
        strl = b'strl' + b'strh' + struct.pack('<I', len(strh)) + strh + b'strf' + struct.pack('<I', len(strf)) + strf
        hdrl = b'hdrl' + b'avih' + struct.pack('<I', len(avih)) + avih + b'LIST' + struct.pack('<I', len(strl)) + strl
        self._f.seek(0)
        self._f.write(b'RIFF' + struct.pack('<I', 0) + b'AVI ')
        self._f.write(b'LIST' + struct.pack('<I', len(hdrl)) + hdrl)
        self._movi_size_at = self._f.tell() + 4
        self._f.write(b'LIST' + struct.pack('<I', 0) + b'movi')

    def write(self, rgb):
        from PIL import Image
        frame = np.asarray(rgb)
        if frame.shape != (self.height, self.width, 3):
            raise ValueError("frame of shape %s does not match the video size %dx%d" % (frame.shape, self.width, self.height))
        buf = io.BytesIO()
        Image.fromarray(np.ascontiguousarray(frame, np.uint8)).save(buf, format='JPEG', quality=self.quality)
        data = buf.getvalue()
        self._f.seek(0, 2)
        at = self._f.tell()
        self._f.write(b'00dc' + struct.pack('<I', len(data)) + data + (b'\0' if len(data) & 1 else b''))
        self._index.append((at - self._movi_fourcc_at, len(data)))
        self._largest = max(self._largest, len(data))
        self.frame_count += 1

    def close(self):
        if self._f is None:
            return
        f = self._f
        f.seek(0, 2)
        movi_end = f.tell()
        f.write(b'idx1' + struct.pack('<I', 16 * len(self._index)))
        for off, size in self._index:
            f.write(b'00dc' + struct.pack('<III', 0x10, off, size))
        end = f.tell()
        self._write_header(self.frame_count)
        f.seek(self._movi_size_at)
        f.write(struct.pack('<I', movi_end - self._movi_fourcc_at))
        f.seek(4)
        f.write(struct.pack('<I', end - 8))
        f.close()
        self._f = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
