"""Host model of the round-off bound behind the grid-culled inside test (csrc/contains.hip, contains_binned_kernel):
``tools/contains_bound_sim.py`` emulates the kernel's fp32 triangle setup, pair test and inflated projected box in numpy and
checks, on pairs placed within a few ulp of the projected triangle borders (slivers, edge-on and tiny triangles, far
offsets), that every pair passing the fp32 test lies inside the box - i.e. culling by the box cannot lose a hit.  The GPU
counterpart (bit-identical hit words against the all-pairs kernel) is tests/test_contains_binned_gpu.py."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))


def test_no_hit_outside_the_inflated_box():
    import contains_bound_sim as sim

    tot_hits = 0
    for seed in (0, 1):
        r = sim.check(seed, T=6000, K=48)
        assert r["lost"] == 0, r
        assert r["boxed"] > 0.7 * 6000 and r["live"] > 0.6 * 6000, r  # the families really are boxed, not all given up on
        assert r["worst_margin_fraction"] < 0.25, r  # the margin has slack: observed excursions stay far inside it
        tot_hits += r["hits"]
    assert tot_hits > 50000  # the border-hugging points do hit


def test_the_t_test_only_removes_hits():
    """The box argument uses the u / v / u+v tests alone; with the t >= tol test on top there are fewer hits, never more."""
    import contains_bound_sim as sim

    a = sim.check(3, T=3000, K=32, need_t=False)
    b = sim.check(3, T=3000, K=32, need_t=True)
    assert b["hits"] <= a["hits"] and b["lost"] == 0
