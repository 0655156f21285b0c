"""The two restatements of the Enum row loop of Column::from_values (striped.rs:340-387) against
each other: the vectorised one the GPU tests use at size and the statement-by-statement one."""
import numpy as np

from oracle import zbra_oracle_np as onp


def test_vectorised_restatement_equals_the_loop():
    rng = np.random.default_rng(21)
    for trial in range(200):
        nv = int(rng.integers(1, 10))
        vt = [int(x) for x in rng.integers(0, 14, nv)]           # duplicate variant tags: the first one wins (:349)
        n = int(rng.integers(0, 300))
        pool = vt + ([99] if trial % 5 == 0 else [])              # sometimes a tag that no variant carries
        tags = rng.choice(pool, n).astype(np.uint32)
        st, variant, rank, counts, bad = onp.enum_rows(tags, vt)
        st2, variant2, rank2, counts2, bad2 = onp.enum_rows_loop(tags, vt)
        assert st == st2 and bad == bad2
        if st == 0:
            assert variant.tolist() == variant2 and rank.tolist() == rank2 and counts.tolist() == counts2
            assert int(counts.sum()) == n


def test_worked_example():
    # rows tagged 7 0 7 1 0 with variants declared in the order 0, 1, 7
    st, variant, rank, counts, _ = onp.enum_rows(np.array([7, 0, 7, 1, 0], dtype=np.uint32), [0, 1, 7])
    assert st == 0 and variant.tolist() == [2, 0, 2, 1, 0] and rank.tolist() == [0, 0, 1, 0, 1] and counts.tolist() == [2, 1, 2]
    st, _, _, _, bad = onp.enum_rows(np.array([0, 1, 5, 9], dtype=np.uint32), [0, 1])
    assert st == 8 and bad == 2
