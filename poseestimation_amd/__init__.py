"""MI355X-native multi-view triangulation: hand-written gfx950 HIP kernels behind a C ABI
(include/mvx.h, poseestimation_amd/libmvx.so) and a drop-in mirror of the reference's `mvn` functions and
model classes for that path (poseestimation_amd.mvn).

    from poseestimation_amd.mvn.utils import op, multiview
    from poseestimation_amd.mvn.models.triangulation import AlgebraicTriangulationNet

`patch_reference()` swaps the kernels into an imported reference `mvn` package instead.
"""
__version__ = "0.1.0"


def patch_reference():
    """Replaces the triangulation functions of an importable reference `mvn` package (its checkout on
    sys.path) with the HIP-backed ones, so that the reference's own model classes and pipelines call the
    kernels unchanged.  The RANSAC model class, the `prepare_batch*` functions and the temporal filter are
    swapped as well when their reference modules import (they pull in optional dependencies such as OpenCV).
    Returns the list of patched attribute names."""
    import importlib

    from poseestimation_amd.mvn.utils import multiview as mv, op as ops

    ref_op = importlib.import_module("mvn.utils.op")
    ref_mv = importlib.import_module("mvn.utils.multiview")
    patched = []
    for name in ("integrate_tensor_2d", "integrate_tensor_3d", "integrate_tensor_3d_with_coordinates",
                 "unproject_heatmaps"):
        setattr(ref_op, name, getattr(ops, name))
        patched.append("mvn.utils.op." + name)
    for name in ("triangulate_batch_of_points", "triangulate_point_from_multiple_views_linear_torch"):
        setattr(ref_mv, name, getattr(mv, name))
        patched.append("mvn.utils.multiview." + name)

    optional = (
        ("mvn.models.triangulation", "poseestimation_amd.mvn.models.triangulation", ("RANSACTriangulationNet",)),
        ("mvn.datasets.utils", "poseestimation_amd.mvn.datasets.utils",
         ("prepare_batch", "prepare_batch_test", "prepare_batch_video")),
        ("savgol_filer", "poseestimation_amd.savgol_filer", ("SAVGOLFilter",)),
    )
    for ref_name, own_name, names in optional:
        try:
            ref_mod = importlib.import_module(ref_name)
        except ImportError:
            continue
        own_mod = importlib.import_module(own_name)
        for name in names:
            setattr(ref_mod, name, getattr(own_mod, name))
            patched.append(ref_name + "." + name)
    return patched
