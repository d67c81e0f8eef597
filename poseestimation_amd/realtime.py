"""The steps the real-time pipelines run on every frame after triangulation (reference:
real_pipeline_trt.py:349-399, pipeline_video.py:399-411): keep the recent poses, smooth them, and turn the
result into the messages sent to the renderer.  The poses stay on the device in a ring of slots; per frame one
launch stores the new pose, smooths the window and maps both frames of a message pair, and one small copy brings
them to the host.  Sockets and queues (send_msg.py) are not part of
this module: `encode` returns the bytes the reference hands to its socket.
"""
import contextlib
import json

import numpy as np
import torch

from poseestimation_amd import _cabi
from poseestimation_amd.savgol_filer import fit_window, hat_matrix

# joint names in wire order and the model joint each one is read from (real_pipeline_trt.py:133-135, :472)
UNITY_JOINTS = ('Hip', 'RightHip', 'RightKnee', 'RightAnkle', 'LeftHip', 'LeftKnee', 'LeftAnkle', 'Spine', 'Thorax',
                'Neck', 'Head', 'LeftShoulder', 'LeftElbow', 'LeftWrist', 'RightShoulder', 'RightElbow', 'RightWrist')
POINT_CHANGE = (6, 2, 1, 0, 3, 4, 5, 7, 8, 16, 9, 13, 14, 15, 12, 11, 10)


_NO_SWITCH = contextlib.nullcontext()


def unity_points(points, scale=1.0, order=POINT_CHANGE):
    """points (..., J, 3) float32 on a HIP device -> joints reordered, axes (x, y, z) -> (-x, z, -y), times scale."""
    p = _cabi.device_f32(points, "points")
    n_joints = p.shape[-2]
    order_t = torch.tensor(order, dtype=torch.int32, device=p.device)
    if order_t.numel() != n_joints or int(order_t.min()) < 0 or int(order_t.max()) >= n_joints:
        raise ValueError("order must list %d joint indices below %d" % (n_joints, n_joints))
    out = torch.empty_like(p)
    with torch.cuda.device(p.device):
        st = _cabi.load().mvx_unity_points(_cabi.ptr(p), _cabi.ptr(order_t), p.numel() // (n_joints * 3), n_joints,
                                           float(scale), _cabi.ptr(out), _cabi.stream(p.device))
    _cabi.check(st, "mvx_unity_points")
    return out


class RealtimeTail:
    """Per-frame state of a real-time pipeline: `push(keypoints_3d)` takes the (J, 3) pose of the new frame (on
    the device) and returns the list of wire messages for it, as the reference builds them: one message until
    more than `slide_window_size` frames have been seen, then two per frame (the mean of the last two smoothed
    poses, then the last smoothed pose; the window is the last `window_frames` = 8 poses, filtered with
    SAVGOLFilter(slide_window_size)).  The scale is 0.43 over the length of the bone between wire joints 2 and 3
    in the first frame (real_pipeline_trt.py:386-392)."""

    def __init__(self, slide_window_size=8, window_frames=8, unity_len=0.43, names=UNITY_JOINTS, order=POINT_CHANGE):
        self.slide_window_size = slide_window_size
        self.window_frames = window_frames
        self.unity_len = unity_len
        self.names = tuple(names)
        self.order = tuple(order)
        self.frames_seen = 0
        self.scale = None
        self._ring = None          # (window_frames, J, 3) on the device: slot frames_seen % window_frames is the oldest
        self._order_t = None
        self._out = None
        self._host = None
        self._host_np = None
        self._ptrs = None
        self._hat = None
        self._window = 1

    def _setup(self, pose):
        n_joints = pose.shape[0]
        dev = pose.device
        self._ring = torch.zeros((self.window_frames, n_joints, 3), dtype=torch.float32, device=dev)
        self._order_t = torch.tensor(self.order, dtype=torch.int32, device=dev)
        self._out = torch.empty((2, n_joints, 3), dtype=torch.float32, device=dev)
        # the two message frames land in page-locked host memory: one asynchronous copy and one stream wait per frame
        self._host = torch.empty((2, n_joints, 3), dtype=torch.float32, pin_memory=True)
        self._host_np = self._host.numpy()
        self._window, order = fit_window(self.slide_window_size, self.window_frames)
        self._hat = hat_matrix(self._window, order, dev)
        self._ptrs = (self._ring.data_ptr(), self._hat.data_ptr(), self._order_t.data_ptr(), self._out.data_ptr())

    def push(self, keypoints_3d):
        """One launch (the pose enters the device ring, the window is smoothed, joints and axes are mapped and
        scaled) and one copy of at most 2 x J x 3 floats to the host per frame."""
        pose = _cabi.device_f32(keypoints_3d, "keypoints_3d").reshape(-1, 3)
        if self._ring is None:
            self._setup(pose)
        if pose.device != self._ring.device:
            _cabi.same_device(self._ring, "the pose ring", keypoints_3d=pose)
        n_joints = pose.shape[0]
        if n_joints != self._ring.shape[1]:
            raise ValueError("pose has %d joints, the first frame had %d" % (n_joints, self._ring.shape[1]))
        slot = self.frames_seen % self.window_frames
        self.frames_seen += 1
        # the reference filters once its frame counter (= poses stored, the new one included) exceeds the window size
        filtered = self.frames_seen > self.slide_window_size and self.frames_seen >= self.window_frames
        first = self.scale is None
        ring, hat, order, out = self._ptrs
        index = pose.device.index
        with (_NO_SWITCH if index == torch.cuda.current_device() else torch.cuda.device(pose.device)):
            st = _cabi.load().mvx_realtime_tail(ring, pose.data_ptr(), slot, self.window_frames, n_joints, hat, self._window,
                                                order, 1.0 if first else float(self.scale), int(filtered), out,
                                                _cabi.raw_stream(index))
            if st != _cabi.MVX_OK:
                _cabi.check(st, "mvx_realtime_tail")
            n_out = 2 if filtered else 1
            self._host[:n_out].copy_(self._out[:n_out], non_blocking=True)
            torch.cuda.current_stream(pose.device).synchronize()
        host = self._host_np[:n_out]
        if first:
            # bone between wire joints 2 and 3 of the first frame; numpy float32 arithmetic as in the pipeline
            self.scale = np.float32(self.unity_len) / np.float32(np.linalg.norm(host[0][2] - host[0][3]))
            host = host * self.scale
        return [self.message(frame) for frame in host]

    def message(self, wire_points):
        """(J, 3) array already in wire order and scale -> {joint name: [x, y, z]} of Python floats."""
        points = np.asarray(wire_points)[:len(self.names)].tolist()   # float32 -> Python float, exactly
        return dict(zip(self.names, points))

    @staticmethod
    def encode(message):
        """The bytes the reference's SendMsg thread writes to its socket (send_msg.py:49-50)."""
        return json.dumps(message).encode('utf-8')
