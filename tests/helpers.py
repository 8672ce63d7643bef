"""Shared helpers for the parity tests (CPU oracle vs HIP path on identical seeded inputs)."""
import numpy as np

from radfoam_amd import foam as foam_mod


def camera_setup(fm, width, height, position=(0.0, 0.0, -3.0)):
    cam = foam_mod.default_camera(width, height)
    cam["position"] = np.asarray(position, dtype=np.float32)
    rays = foam_mod.camera_rays(cam)
    start = foam_mod.nearest_point(fm["points"], cam["position"])
    return cam, rays, np.uint32(start)


def random_rays(fm, n_rays, seed):
    """Incoherent rays from a few origins inside/outside the foam, with their entry cells."""
    rng = np.random.default_rng(seed)
    origins = np.array([[0.0, 0.0, -3.0], [2.5, 0.3, 0.2], [0.05, -0.02, 0.01], [-0.4, 0.5, 0.3]], dtype=np.float32)
    starts = np.array([foam_mod.nearest_point(fm["points"], o) for o in origins], dtype=np.uint32)
    which = rng.integers(0, len(origins), size=n_rays)
    target = rng.uniform(-0.7, 0.7, size=(n_rays, 3)).astype(np.float32)
    d = target - origins[which]
    d = d * rng.uniform(0.5, 2.0, size=(n_rays, 1)).astype(np.float32)  # un-normalised on purpose
    rays = np.concatenate([origins[which], d.astype(np.float32)], axis=1).astype(np.float32)
    return rays, starts[which]


def to_torch_foam(fm, device, attr_dtype=None):
    import torch

    attrs = torch.from_numpy(fm["attributes"])
    if attr_dtype is not None:
        attrs = attrs.to(attr_dtype)
    return (
        torch.from_numpy(fm["points"]).to(device),
        attrs.to(device),
        torch.from_numpy(fm["point_adjacency"]).to(device),
        torch.from_numpy(fm["point_adjacency_offsets"]).to(device),
    )


def grad_close(got, ref, rtol=1e-3):
    """Gradient parity bound of the north star (1e-3 relative): every element within
    rtol*|ref| + rtol*rms(ref) (the second term absorbs summation-order noise on elements that
    are sums of cancelling contributions); also returns the global relative L2 error."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    nz = ref[ref != 0]
    rms = np.sqrt(np.mean(nz ** 2)) if nz.size else 0.0
    err = np.abs(got - ref)
    bound = rtol * np.abs(ref) + rtol * rms
    rel_l2 = np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-30)
    return bool((err <= bound).all()), float(rel_l2), float((err / np.maximum(bound, 1e-30)).max())
