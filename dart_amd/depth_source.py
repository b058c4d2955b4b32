"""ImageDepthSource<ushort, uchar3> (src/depth_sources/image_depth_source.h:27-102): a recorded sequence of 16-bit PNG depth
frames, found by directory listing (sorted, :153-208) or by a printf-style file name format (:345-420), feeding the
depth -> vertex -> normal front end (PointCloudSource::projectDepthMap, point_cloud_src.h:261-285).

Colour frames and JPG depth are outside the hot path and not read."""
import os

import numpy as np

from .optimizer import depth_to_vertices, vertices_to_normals


class ImageDepthSource:
    def __init__(self):
        self._files = []
        self._frame = 0
        self._depth = None
        self._png_swap = True           # image_depth_source.h:141
        self.scale_to_meters = 1.0
        self.focal_length = (1.0, 1.0)
        self.principal_point = (0.0, 0.0)
        self.width = self.height = 0

    # ---- the two initialisers
    def initialize_directory(self, depth_directory, focal_length, principal_point=(0, 0), depth_width=0, depth_height=0, scale_to_meters=1.0):
        """Directory variant: every *.png of the directory in sorted order (:176-208). Width/height 0 = take the first frame's."""
        if not os.path.isdir(depth_directory):
            return False
        files = sorted(os.path.join(depth_directory, f) for f in os.listdir(depth_directory) if f.lower().endswith(".png"))
        return self._finish(files, focal_length, principal_point, depth_width, depth_height, scale_to_meters)

    def initialize_format(self, depth_filename_format, focal_length, principal_point=(0, 0), depth_width=0, depth_height=0, first_depth_frame=0,
                          scale_to_meters=1.0):
        """printf variant: frames first, first+1, ... while the file exists (:345-420)."""
        files = []
        k = first_depth_frame
        while os.path.exists(depth_filename_format % k):
            files.append(depth_filename_format % k)
            k += 1
        return self._finish(files, focal_length, principal_point, depth_width, depth_height, scale_to_meters)

    def _finish(self, files, focal_length, principal_point, w, h, scale):
        if not files:
            return False
        self._files = files
        self.focal_length = (float(focal_length[0]), float(focal_length[1]))
        self.scale_to_meters = float(scale)
        if w == 0 or h == 0:
            from PIL import Image
            with Image.open(files[0]) as im:
                w, h = im.size
        self.width, self.height = int(w), int(h)
        pp = (float(principal_point[0]), float(principal_point[1]))
        # only x is tested, and the centre is an INTEGER division (image_depth_source.h:227-231, :402-406)
        self.principal_point = pp if pp[0] != 0.0 else (float(self.width // 2), float(self.height // 2))
        self._frame = 0
        return self._read()

    # ---- DepthSource interface
    def setPngSwap(self, swap):
        self._png_swap = bool(swap)

    def getNumDepthFrames(self):
        return len(self._files)

    def getFrame(self):
        return self._frame

    def getDepthWidth(self):
        return self.width

    def getDepthHeight(self):
        return self.height

    def getScaleToMeters(self):
        return self.scale_to_meters

    def setFrame(self, frame):
        self._frame = int(frame)
        return self._read()

    def advance(self):
        self._frame += 1
        if self._frame == len(self._files):     # wraps around (:503-506)
            self._frame = 0
        return self._read()

    def getDepth(self):
        """The current frame, uint16 [height, width] (raw sensor units)."""
        return self._depth

    def _read(self):
        from PIL import Image
        with Image.open(self._files[self._frame]) as im:
            if im.size != (self.width, self.height):
                raise ValueError("expected %dx%d depth but got %dx%d" % (self.width, self.height, im.size[0], im.size[1]))
            if im.mode not in ("I;16", "I;16B", "I;16L", "I"):
                raise ValueError("expected 16 bits per pixel, got mode %s" % im.mode)
            a = np.asarray(im)
        a = a.astype(np.uint16)
        if not self._png_swap:      # libpng hands back big-endian samples unless png_set_swap is on (:614-616)
            a = a.byteswap()
        self._depth = np.ascontiguousarray(a)
        return True

    # ---- the front end of the hot path
    def device_depth(self, device=None):
        """depth in metres on the GPU: (float)raw * scaleToMeters, as depthToVertices<ushort> scales it"""
        import torch
        dev = device or torch.device("cuda", torch.cuda.current_device())
        d = self._depth.astype(np.float32) * np.float32(self.scale_to_meters)
        return torch.from_numpy(d).to(dev)

    def vertices_and_normals(self, depth_range=(0.1, 5.0), device=None):
        depth = self.device_depth(device)
        vert = depth_to_vertices(depth, self.principal_point, self.focal_length, depth_range)
        return depth, vert, vertices_to_normals(vert)
