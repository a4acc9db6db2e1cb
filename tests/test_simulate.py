"""The read / genome simulator's `grch38_like` profile (bwams/simulate.py): seeded, and it really holds the structures it names."""
import numpy as np

from bwams import simulate


def test_grch38_like_is_seeded_and_structured():
    n = 6_000_000
    g, holes = simulate.make_genome(n, seed=5, profile="grch38_like", return_holes=True)
    g2 = simulate.make_genome(n, seed=5, profile="grch38_like")
    assert g.dtype == np.uint8 and len(g) == n and int(g.max()) <= 3 and np.array_equal(g, g2)
    assert not np.array_equal(g, simulate.make_genome(n, seed=6, profile="grch38_like"))
    assert holes.dtype == simulate.HOLE_DTYPE and len(holes) >= 2 and np.all(np.diff(holes["offset"]) >= 0)
    big = holes[np.argmax(holes["len"])]
    assert big["len"] == n // 200                                      # the centromere-like hole beside the satellite array ...
    # ... which ends where the hole begins: 171-bp monomers in 12-monomer higher-order repeats, the first 20 HOR copies exact
    arr_end = int(big["offset"])
    n_mono = min(10_000, max(48, n // (171 * 60)))
    hor = 12 * 171
    arr = g[arr_end - (n_mono // 12) * hor:arr_end]
    assert np.array_equal(arr[:hor], arr[hor:2 * hor]) and np.array_equal(arr[:hor], arr[19 * hor:20 * hor])
    later = arr[40 * hor:41 * hor]
    assert 0 < (later != arr[:hor]).mean() < 0.03                       # HOR copies 0.5 % apart (twice that between two mutated ones)
    mono = arr[:hor].reshape(12, 171)
    assert 0.15 < (mono[0] != mono[1]).mean() < 0.5                     # monomers of one HOR ~ 20 % from the consensus each
    # exact microsatellites and poly-A / poly-T runs
    s = bytes(g)
    assert s.count(bytes([1, 0] * 20)) > 5 and s.count(bytes([2, 0, 3, 0] * 10)) > 5           # (CA)n, (GATA)n
    assert s.count(bytes([0] * 15)) + s.count(bytes([3] * 15)) > n // 40_000
    # exact segmental duplications of 150 kb: 64-mers sampled every ~49 kb that occur a second time (a random 64-mer does not; the
    # microsatellites and the satellite array's exact head add a few more)
    seg = min(150_000, n // 40)
    probe, step = 64, 49_157
    twice = 0
    for p in range(17, n - probe, step):
        k = s[p:p + probe]
        if s.find(k) != p or s.find(k, p + 1) != -1:
            twice += 1
    assert twice >= 3 * 2 * seg // step // 2                             # three direct copies and their sources; half of what is expected
    # the small genome still works (everything scaled down)
    small, h2 = simulate.make_genome(100_000, seed=3, profile="grch38_like", return_holes=True)
    assert len(small) == 100_000 and len(h2) >= 2
