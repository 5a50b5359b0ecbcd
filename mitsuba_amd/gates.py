"""Stated bounds of the image-level correctness gates. The tests assert them (tests/bidir_pin.py, tests/test_gpu_parity.py,
tests/test_oracle.py); bench.py prints them beside every throughput figure.

REL_MSE_ALL_CELLS / REL_MSE_LIT_CELLS: relative mean squared error, mean(((x - ref) / ref)^2), of 48-pixel cell luminances
of a `path` render of the Veach room against testScene/veach_bidir/bidir.exr, the one rendered image the reference
holds. That file is a photonmapper render that carries direct light twice (tests/bidir_pin.py), so the render enters as
2 D + I (D: emitted + direct, I: the rest); over every cell outside the banner, and -- with the centre constant 0.96
applied -- over the directly lit cells. Measured: GPU at 768x576 and 4 096 samples 0.079 and 0.00032; CPU restatement at
192x144 and 768 samples 0.088 and 0.00044. A plain render (direct light once) scores 0.17 over the lit cells.

SEED_NOISE_FACTOR: at matched samples per pixel, relMSE(GPU render, CPU restatement render with another seed) <=
SEED_NOISE_FACTOR x relMSE(two CPU restatement renders with different seeds), on the Cornell box and the Veach room
(measured: 0.99x and 0.2-1.0x). SAME_SEED_FACTOR: with the same seed the two share their sample streams and differ by
less than this fraction of the seed noise (measured: 5e-8 and 1e-7)."""

REL_MSE_ALL_CELLS = 0.10
REL_MSE_LIT_CELLS = 0.001
SEED_NOISE_FACTOR = 2.0
SAME_SEED_FACTOR = 0.02


def as_dict():
    return {"rel_mse_vs_reference_image_all_cells_max": REL_MSE_ALL_CELLS, "rel_mse_vs_reference_image_lit_cells_max": REL_MSE_LIT_CELLS,
            "rel_mse_gpu_vs_cpu_over_seed_noise_max": SEED_NOISE_FACTOR, "rel_mse_same_seed_over_seed_noise_max": SAME_SEED_FACTOR,
            "measured": {"rel_mse_vs_reference_image_all_cells": 0.079, "rel_mse_vs_reference_image_lit_cells": 0.00032,
                         "gpu_vs_cpu_over_seed_noise": {"cornell": 0.99, "veach": 0.2}},
            "tests": ["tests/test_gpu_parity.py::test_veach_room_against_the_reference_held_image",
                      "tests/test_gpu_parity.py::test_image_rel_mse_is_within_seed_noise"]}
