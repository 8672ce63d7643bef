"""-m gpu: the operators compose into an optimisation loop (examples/fit_synthetic.py): fused attribute
packing -> tracer forward -> loss -> tracer backward -> packing backward -> Adam, and the loss goes down."""
import pytest

pytestmark = pytest.mark.gpu


def test_fit_loop_reduces_the_loss():
    from examples import fit_synthetic

    losses = fit_synthetic.fit(num_points=8000, sh_degree=1, width=96, height=72, steps=40, log=lambda *_: None)
    assert all(l == l for l in losses)            # finite
    assert losses[-1] < 0.35 * losses[0], (losses[0], losses[-1])


def test_fit_loop_with_moving_points_and_gpu_rebuilds():
    """examples/fit_points.py: positions are optimised too and the triangulation follows them on the GPU
    (Triangulation.rebuild(points, incremental=True), scene.py:160-200); the lists the loop ends with are the
    Delaunay lists of the points it ends with (Qhull)."""
    import numpy as np

    from examples import fit_points
    from radfoam_amd import foam

    r = fit_points.fit(num_points=8000, sh_degree=1, width=96, height=72, steps=40, rebuild_every=5,
                       log=lambda *_: None)
    losses = r["losses"]
    assert all(l == l for l in losses)
    assert losses[-1] < 0.8 * losses[0], (losses[0], losses[-1])
    assert len(r["rebuild_ms"]) == 8
    off, adj = foam.delaunay_csr(r["points"].cpu().numpy())
    assert np.array_equal(r["offsets"].cpu().numpy(), off) and np.array_equal(r["adjacency"].cpu().numpy(), adj)
