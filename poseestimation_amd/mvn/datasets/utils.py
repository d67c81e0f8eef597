"""Same functions as the reference's mvn/datasets/utils.py: the collate functions of its data loaders and
the `prepare_batch*` step that turns a collated batch into the tensors the triangulation models take.

The reference uploads one small tensor per view and per camera; here each kind of data crosses to the
device once: the images as one array, the projection matrices as one (B, V, 3, 4) float32 array built on
the host in float64 from `Camera.projection`, exactly as the reference builds each of them.
"""
import numpy as np
import torch


def _choose_views(items, randomize_n_views, min_n_views, max_n_views):
    """View indices shared by every item of the batch; consumes numpy's global RNG as the reference does
    (one `randint`, then one `choice` without replacement).  Reference: mvn/datasets/utils.py:15-22."""
    total_n_views = min(len(item["images"]) for item in items)
    if not randomize_n_views:
        return np.arange(total_n_views)
    n_views = np.random.randint(min_n_views, min(total_n_views, max_n_views) + 1)
    return np.random.choice(np.arange(total_n_views), size=n_views, replace=False)


def _collate(items, randomize_n_views, min_n_views, max_n_views, with_ground_truth):
    items = [item for item in items if item is not None]
    if not items:
        print("All items in batch are None")
        return None
    views = _choose_views(items, randomize_n_views, min_n_views, max_n_views)
    batch = {
        # (B, V, ...): item-major, the chosen views in the chosen order
        "images": np.stack([np.stack([item["images"][v] for v in views], axis=0) for item in items], axis=0),
        "detections": np.array([[item["detections"][v] for v in views] for item in items]),
        # [V][B], the layout the models index as batch['cameras'][view][frame]
        "cameras": [[item["cameras"][v] for item in items] for v in views],
    }
    if with_ground_truth:
        batch["keypoints_3d"] = [item["keypoints_3d"] for item in items]
    batch["indexes"] = [item["indexes"] for item in items]
    if with_ground_truth:
        batch["actname"] = [item["actname"] for item in items]
        try:
            batch["pred_keypoints_3d"] = np.array([item["pred_keypoints_3d"] for item in items])
        except Exception:   # items without predictions: the key stays absent, as in the reference
            pass
    return batch


def make_collate_fn(randomize_n_views=True, min_n_views=10, max_n_views=31):
    """Training/validation collate.  Reference: mvn/datasets/utils.py:6-40."""
    def collate_fn(items):
        return _collate(items, randomize_n_views, min_n_views, max_n_views, True)
    return collate_fn


def make_collate_fn_test(randomize_n_views=True, min_n_views=10, max_n_views=31):
    """Collate for data without ground truth.  Reference: mvn/datasets/utils.py:42-68."""
    def collate_fn(items):
        return _collate(items, randomize_n_views, min_n_views, max_n_views, False)
    return collate_fn


def worker_init_fn(worker_id):
    """Distinct numpy seed per loader worker.  Reference: mvn/datasets/utils.py:71-72."""
    np.random.seed(np.random.get_state()[1][0] + worker_id)


def stack_projections(cameras, device):
    """cameras[view][frame] -> (B, V, 3, 4) float32 on `device`.  Each matrix is K.dot([R|t]) in float64
    (`Camera.projection`), rounded to float32 once; one host array, one upload.
    Reference: mvn/datasets/utils.py:91-92."""
    n_views, n_frames = len(cameras), len(cameras[0])
    host = np.empty((n_frames, n_views, 3, 4), dtype=np.float64)
    for v, per_frame in enumerate(cameras):
        for b, camera in enumerate(per_frame):
            host[b, v] = camera.projection
    return torch.from_numpy(host.astype(np.float32)).to(device)


def _images_to_device(images, device):
    """(B, V, H, W, C) array -> (B, V, C, H, W) float32 on `device`: one upload in the array's own dtype
    (float64 is rounded to float32 on the host first: same values, half the bytes), layout change and cast
    on the device.  Reference: mvn/datasets/utils.py:76-82 with mvn/utils/img.py:98-101."""
    images = np.asarray(images)
    if images.dtype == np.float64:
        images = images.astype(np.float32)
    return torch.from_numpy(np.ascontiguousarray(images)).to(device).permute(0, 1, 4, 2, 3).float().contiguous()


def _keypoints_gt(batch, device):
    stacked = np.stack(batch["keypoints_3d"], axis=0)
    both = torch.from_numpy(np.ascontiguousarray(stacked)).float().to(device)
    return both[:, :, :3].contiguous(), both[:, :, 3:].contiguous()


def prepare_batch(batch, device, config, is_train=True):
    """-> (images (B,V,C,H,W), keypoints_3d_gt (B,J,3), keypoints_3d_validity_gt (B,J,1), proj (B,V,3,4)).
    Reference: mvn/datasets/utils.py:74-94."""
    images = _images_to_device(batch["images"], device)
    keypoints, validity = _keypoints_gt(batch, device)
    return images, keypoints, validity, stack_projections(batch["cameras"], device)


def prepare_batch_test(batch, device, config, is_train=True):
    """-> (images, proj).  Reference: mvn/datasets/utils.py:97-111."""
    return _images_to_device(batch["images"], device), stack_projections(batch["cameras"], device)


def prepare_batch_video(sample, views, device):
    """One frame of the real-time pipelines: sample['inps'] is a list of `views` image tensors (C, H, W),
    sample['cameras'] a list of cameras -> (images (1,V,C,H,W), proj (1,V,3,4)).
    Reference: mvn/datasets/utils.py:113-136 (which only ever fills one frame: index // views stays 0 for
    the `views` images it is given; more images than `views` raise IndexError there and ValueError here)."""
    images = list(sample["inps"])
    if len(images) > views and views > 0:
        raise ValueError("prepare_batch_video takes one frame: %d images for %d views" % (len(images), views))
    images_batch = torch.stack(images, dim=0).unsqueeze(0).to(device)
    proj = stack_projections([[camera] for camera in sample["cameras"]], device)
    return images_batch, proj
