"""Same classes and functions as the reference's mvn/utils/volumetric.py: the cuboid container, the
Euler-Rodrigues rotation matrix and the rotation of a coordinate volume.  Rendering needs OpenCV, which is
imported only when a `render` method is called."""
import numpy as np
import torch

from poseestimation_amd.mvn.utils import multiview


def _cv2():
    import cv2  # noqa: deferred: only the render methods need it
    return cv2


class Point3D:
    """A 3-D point drawn as a filled circle.  Reference: mvn/utils/volumetric.py:8-22."""

    def __init__(self, point, size=3, color=(0, 0, 255)):
        self.point = point
        self.size = size
        self.color = color

    def render(self, proj_matrix, canvas):
        p = multiview.project_3d_points_to_image_plane_without_distortion(proj_matrix, np.array([self.point]))[0]
        _cv2().circle(canvas, tuple(int(c) for c in p), self.size, self.color, self.size)
        return canvas


class Line3D:
    """A 3-D segment drawn as a line.  Reference: mvn/utils/volumetric.py:25-41."""

    def __init__(self, start_point, end_point, size=2, color=(0, 0, 255)):
        self.start_point, self.end_point = start_point, end_point
        self.size = size
        self.color = color

    def render(self, proj_matrix, canvas):
        a, b = multiview.project_3d_points_to_image_plane_without_distortion(
            proj_matrix, np.array([self.start_point, self.end_point]))
        _cv2().line(canvas, tuple(int(c) for c in a), tuple(int(c) for c in b), self.color, self.size)
        return canvas


class Cuboid3D:
    """Axis-aligned box given by its minimum corner `position` and edge lengths `sides`.
    Reference: mvn/utils/volumetric.py:44-84."""

    def __init__(self, position, sides):
        self.position = position
        self.sides = sides

    def build(self):
        """The 12 edges as Line3D: the three edges at the minimum corner are coloured by axis (x red, y green,
        z blue in BGR order as the reference draws them), the rest share one colour."""
        sx, sy, sz = self.sides[0], self.sides[1], self.sides[2]
        rest = (255, 255, 0)
        corners_and_edges = [
            (np.array([0, 0, 0]), [((sx, 0, 0), (255, 0, 0)), ((0, sy, 0), (0, 255, 0)), ((0, 0, sz), (0, 0, 255))]),
            (np.array([sx, 0, sz]), [((-sx, 0, 0), rest), ((0, sy, 0), rest), ((0, 0, -sz), rest)]),
            (np.array([sx, sy, 0]), [((-sx, 0, 0), rest), ((0, -sy, 0), rest), ((0, 0, sz), rest)]),
            (np.array([0, sy, sz]), [((sx, 0, 0), rest), ((0, -sy, 0), rest), ((0, 0, -sz), rest)]),
        ]
        lines = []
        for corner, edges in corners_and_edges:
            start = self.position + corner
            for delta, color in edges:
                lines.append(Line3D(start, start + np.array(delta), color=color))
        return lines

    def render(self, proj_matrix, canvas):
        for line in self.build():
            canvas = line.render(proj_matrix, canvas)
        return canvas


def get_rotation_matrix(axis, theta):
    """Rotation by theta radians about `axis` from the Euler-Rodrigues parameters
    (a, b, c, d) = (cos(theta/2), -axis*sin(theta/2)).  Reference: mvn/utils/volumetric.py:87-99."""
    axis = np.asarray(axis)
    axis = axis / np.sqrt(np.dot(axis, axis))
    a = np.cos(theta / 2.0)
    b, c, d = -axis * np.sin(theta / 2.0)
    aa, bb, cc, dd = a * a, b * b, c * c, d * d
    bc, ad, ac, ab, bd, cd = b * c, a * d, a * c, a * b, b * d, c * d
    return np.array([[aa + bb - cc - dd, 2 * (bc + ad), 2 * (bd - ac)],
                     [2 * (bc - ad), aa + cc - bb - dd, 2 * (cd + ab)],
                     [2 * (bd + ac), 2 * (cd - ab), aa + dd - bb - cc]])


def rotate_coord_volume(coord_volume, theta, axis):
    """Applies get_rotation_matrix(axis, theta), cast to float32, to every (x, y, z) of a (..., 3) volume.
    Reference: mvn/utils/volumetric.py:102-114."""
    rot = torch.from_numpy(get_rotation_matrix(axis, theta)).type(torch.float).to(coord_volume.device)
    flat = coord_volume.reshape(-1, 3)
    return rot.mm(flat.t()).t().reshape(coord_volume.shape)
