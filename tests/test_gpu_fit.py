"""-m gpu: the operators compose into an optimisation loop (examples/fit_synthetic.py): fused attribute
packing -> tracer forward -> loss -> tracer backward -> packing backward -> Adam, and the loss goes down."""
import pytest

pytestmark = pytest.mark.gpu


def test_fit_loop_reduces_the_loss():
    from examples import fit_synthetic

    losses = fit_synthetic.fit(num_points=8000, sh_degree=1, width=96, height=72, steps=40, log=lambda *_: None)
    assert all(l == l for l in losses)            # finite
    assert losses[-1] < 0.35 * losses[0], (losses[0], losses[-1])
