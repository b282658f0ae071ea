"""Pin the CPU oracle (oracle/mic_oracle.c) against the fixtures captured from the reference.

The fixtures were produced by tests/golden/make_golden.py importing /root/reference
(compositor.composite, background_resizing, macro_placement_test) on Pillow 12.2.0.
Bar: bit-exact (the path is 8-bit integer arithmetic).
"""
import json
import os

import numpy as np
import pytest

import cases
import oracle


def _load(golden_dir, stem):
    with open(os.path.join(golden_dir, stem + ".json"), encoding="utf-8") as f:
        meta = json.load(f)
    npz_path = os.path.join(golden_dir, stem + ".npz")
    arrays = np.load(npz_path) if os.path.exists(npz_path) else None
    return meta, arrays


def test_composite_cases_bit_exact(golden_dir):
    meta, arrays = _load(golden_dir, "composite")
    all_cases = {c["name"]: c for c in cases.composite_cases()}
    assert len(meta["cases"]) == len(all_cases)
    for row in meta["cases"]:
        c = all_cases[row["name"]]
        got = oracle.composite(c["bg"], c["objects"], c["placements"])
        want = arrays[row["name"]]
        assert got.shape == want.shape, row["name"]
        assert np.array_equal(got, want), (row["name"], int(np.abs(got.astype(int) - want).max()))
        assert cases.sha16(got) == row["sha16"]


def test_reference_own_test_assertion(golden_dir):
    """tests/test_compositor.py:5-11 of the reference, restated on the oracle."""
    c = cases.composite_kat_cases()[0]
    out = oracle.composite(c["bg"], c["objects"], c["placements"])
    assert tuple(out[4, 4, :3]) == (0, 255, 0)


def test_known_answers_appendix_a6():
    k = {c["name"]: c for c in cases.composite_kat_cases()}
    red = np.array([255, 0, 0, 255])

    def changed(name):
        c = k[name]
        out = oracle.composite(c["bg"], c["objects"], c["placements"])
        ys, xs = np.nonzero((out != c["bg"]).any(axis=2))
        return out, sorted(set(zip(xs.tolist(), ys.tolist())))

    out, px = changed("clip_bottom_right")
    assert px == [(8, 8), (8, 9), (9, 8), (9, 9)] and tuple(out[8, 8]) == (127, 128, 0, 255)
    out, px = changed("clip_top_left")
    assert px == [(0, 0), (0, 1), (1, 0), (1, 1)]
    assert len(changed("degenerate_zero")[1]) == 1 and len(changed("degenerate_negative")[1]) == 1
    out, px = changed("float_box_str_id")
    assert {p[0] for p in px} == {1, 2, 3, 4} and {p[1] for p in px} == {1, 2, 3, 4}
    out, _ = changed("semi_transparent_bg")
    assert tuple(out[3, 3]) == (3, 189, 8, 178)
    assert (out[0, 0] == [10, 20, 30, 100]).all() and (red == k["clip_top_left"]["bg"][5, 5]).all()


def test_resize_cases_bit_exact(golden_dir):
    meta, arrays = _load(golden_dir, "resize")
    for i, row in enumerate(meta["cases"]):
        c = cases.resize_case(i)
        assert c["name"] == row["name"]
        got = oracle.resize(c["src"], c["size"], oracle.LANCZOS)
        assert np.array_equal(got, arrays[row["name"]]), row["name"]
        gotb = oracle.resize(c["src"], c["size"], oracle.BILINEAR)
        assert np.array_equal(gotb, arrays[row["name"] + "_bilinear"]), row["name"] + " bilinear"


def test_median_cases(golden_dir):
    meta, _ = _load(golden_dir, "median")
    rows = {r["name"]: r for r in meta["cases"]}
    for i in range(cases.N_MEDIAN):
        c = cases.median_case(i)
        assert list(oracle.median_rgb(c["rgba"])) == rows[c["name"]]["rgb"], c["name"]


def _load_bundle(bundle):
    from PIL import Image  # PNG decode only
    base = os.path.join(cases.BUNDLE_DIR, bundle)
    with open(os.path.join(base, "results.json"), encoding="utf-8") as f:
        items = json.load(f)
    objs = {int(it["object_id"]): np.array(Image.open(os.path.join(base, it["filename"])).convert("RGBA"))
            for it in items}
    bg = np.array(Image.open(os.path.join(base, "background.png")).convert("RGBA"))
    return objs, bg


def test_bundle_medians_and_known_hashes(golden_dir):
    pytest.importorskip("PIL")
    meta, _ = _load(golden_dir, "median")
    rows = {r["name"]: r for r in meta["cases"]}
    assert rows["bundle_squarespace"]["rgb"] == [220, 238, 245]
    assert rows["bundle_audio_book"]["rgb"] == [38, 73, 115]
    for b in cases.BUNDLES:
        _, bg = _load_bundle(b)
        assert list(oracle.median_rgb(bg)) == rows[f"bundle_{b}"]["rgb"]
        assert int((bg[:, :, 3] > 0).sum()) == rows[f"bundle_{b}"]["nontransparent"]


def test_bundle_composites_c1(golden_dir):
    """BASELINE.json configs[0] (squarespace 1:1) and the App. A.6 table, via oracle + goldens."""
    pytest.importorskip("PIL")
    meta, arrays = _load(golden_dir, "bundles")
    survey_sha = {"squarespace_1x1": "f558bc6442779fe2", "squarespace_9x16": "3e495c91a4e7ae89",
                  "squarespace_16x9": "794dd56e07244e43", "squarespace_21x9": "52670da899b779c4",
                  "audio_book_1x1": "b46c3ec186250031", "audio_book_9x16": "0fb27dc72c429aee",
                  "audio_book_16x9": "ce806dcab2621dfe", "audio_book_21x9": "f1bd6992be978124",
                  "squarespace_resample_kat": "304fc6ba0a718e29"}
    bundles = {b: _load_bundle(b) for b in cases.BUNDLES}
    for row in meta["cases"]:
        if row["name"] in survey_sha:
            assert row["sha16"] == survey_sha[row["name"]]
        objs, bgimg = bundles[row["bundle"]]
        W, H = row["canvas"]
        bg = oracle.fill_solid((W, H), oracle.median_rgb(bgimg) + (255,))
        if "placements" in row:
            pl = row["placements"]
        else:
            ids = [c["object_id"] for c in row["layout"]["root"]["children"]]
            pl = [{"object_id": i, "box": bx} for i, bx in zip(ids, row["boxes"])]
        got = oracle.composite(bg, objs, pl)
        assert cases.sha16(got) == row["sha16"], row["name"]
        if row["name"] in arrays.files:
            assert np.array_equal(got, arrays[row["name"]])


def test_thumbnails_and_size_rule(golden_dir):
    pytest.importorskip("PIL")
    meta, arrays = _load(golden_dir, "contact_sheet")
    for r in meta["thumbnail_sizes"]:
        assert list(oracle.thumbnail_size(r["src"])) == r["size"], r
    for row in meta["cases"]:
        if row.get("bundle") is None:
            big = cases.synthetic.make_cutout(np.random.default_rng(50_001), 1000, 800, "soft")
            th = oracle.thumbnail(big)
            assert list(th.shape[1::-1]) == row["size"] and cases.sha16(th) == row["sha16"]
            continue
        objs, _ = _load_bundle(row["bundle"])
        for t in row["thumbs"]:
            th = oracle.thumbnail(objs[t["object_id"]])
            assert np.array_equal(th, arrays[f"{row['bundle']}_thumb_{t['object_id']}"])


def test_overlay_rectangles_and_candidates_grid(golden_dir):
    """_save_overlay_debug and _compose_candidates_grid of the reference
    (macro_placement_test.py:967-983, 1332-1345), pixels captured by make_golden.py."""
    with open(os.path.join(golden_dir, "overlay.json"), encoding="utf-8") as f:
        meta = json.load(f)
    arrays = np.load(os.path.join(golden_dir, "overlay.npz"))
    seen = 0
    for i in range(cases.N_OVERLAY):
        c = cases.overlay_case(i)
        got = oracle.overlay_debug(c["placements"], c["canvas"])
        assert np.array_equal(got, arrays[c["name"]]), c["name"]
        seen += 1
    for i in range(cases.N_GRID):
        c = cases.grid_case(i)
        got = oracle.candidates_grid(c["images"])
        assert np.array_equal(got, arrays[c["name"]]), c["name"]
        seen += 1
    assert seen == len(meta["cases"])
