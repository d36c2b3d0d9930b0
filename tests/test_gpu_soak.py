"""In-suite soak: the randomised sequences of tests/test_gpu_state_machine.py on seeds NOBODY picked — the base
changes from run to run (tests/conftest.py: SOAK_BASE, printed in the report header; OMX_SOAK_SEED=<base> reproduces a run).  60
sequences per run: six seeds for each of ten cases — the nine of tools/soak_ragged.py (which remains the long-running form of the same
thing: 4 950 sequences in round 3).  Green here means the bars hold by RULE — conditioning-derived for the reassigned columns
(parity.conditioned_bar), the half-integer rule for the oscilloscope's integer geometry — not by choice of seed."""
import pytest

import conftest
import test_gpu_parity_meters as m
import test_gpu_state_machine as t
from openmeters_amd import capi

pytestmark = pytest.mark.gpu
N = 6


@pytest.mark.parametrize("seed", conftest.soak_seeds(N, 0))
def test_soak_spectrogram_operation_sequences(omx, oracle, seed):
    t.test_spectrogram_random_operation_sequences(omx, oracle, seed)


@pytest.mark.parametrize("seed", conftest.soak_seeds(N, 1))
def test_soak_oscilloscope_operation_sequences(omx, oracle, seed):
    t.test_oscilloscope_random_operation_sequences(omx, oracle, seed)


@pytest.mark.parametrize("seed", conftest.soak_seeds(N, 2))
def test_soak_meter_block_sequences(omx, oracle, seed):
    t.test_meter_processors_random_block_sequences(omx, oracle, seed)


@pytest.mark.parametrize("seed", conftest.soak_seeds(N, 3))
def test_soak_ragged_loudness_chunk_parallel(omx, oracle, seed):
    k = seed % 4
    t.test_ragged_loudness_bank_chunk_parallel_form_matches_per_stream_oracles(omx, oracle, seed, (2, 6, 8, 3)[k], (48000.0, 44100.0, 48000.0, 96000.0)[k],
                                                                               (256, 128, 64, 512)[k])


@pytest.mark.parametrize("seed", conftest.soak_seeds(N, 4))
def test_soak_ragged_loudness_sequential(omx, oracle, seed):
    k = seed % 5
    t.test_ragged_loudness_bank_random_per_stream_block_counts_match_per_stream_oracles(
        omx, oracle, seed, (2, 8, 6, 1, 3)[k], (48000.0, 48000.0, 96000.0, 44100.0, 192000.0)[k], (256, 256, 100, 37, 64)[k])


@pytest.mark.parametrize("seed", conftest.soak_seeds(N, 5))
def test_soak_ragged_stereometer(omx, oracle, seed):
    if seed % 2:   # chunk-parallel form, 2 channels
        t.test_ragged_stereometer_bank_random_per_stream_block_counts_match_per_stream_oracles(omx, oracle, seed, 2, seed % 3 != 2, seed % 3 == 0, 2)
    else:          # sequential form
        t.test_ragged_stereometer_bank_random_per_stream_block_counts_match_per_stream_oracles(omx, oracle, seed, 2 if seed % 4 else 6, True, True, 1)


@pytest.mark.parametrize("seed", conftest.soak_seeds(N, 6))
def test_soak_ragged_waveform(omx, oracle, seed):
    t.test_ragged_waveform_bank_random_per_stream_frame_counts_match_per_stream_oracles(omx, oracle, seed, 2 if seed % 3 else 6, seed % 2 == 0,
                                                                                          (48000.0, 44100.0, 8000.0)[seed % 3])


@pytest.mark.parametrize("seed", conftest.soak_seeds(N, 7))
def test_soak_ragged_oscilloscope(omx, oracle, seed):
    t.test_ragged_oscilloscope_bank_random_per_stream_block_counts_match_per_stream_processors(
        omx, oracle, seed, capi.TRIGGER_ZERO_CROSSING if seed % 3 == 2 else capi.TRIGGER_STABLE)


@pytest.mark.parametrize("seed", conftest.soak_seeds(N, 8))
def test_soak_ragged_spectrogram_bank(omx, oracle, seed):
    W, hop, reassign = ((1024, 256, True), (4096, 256, True), (2048, 64, True), (1024, 300, False), (2048, 777, True), (4096, 100, True))[seed % 6]
    t.test_ragged_bank_random_per_stream_op_sequences_match_per_stream_oracles(omx, oracle, seed, W, hop, reassign)


@pytest.mark.parametrize("seed", conftest.soak_seeds(N, 9))
def test_soak_waveform_chunk_parallel(omx, oracle, seed):
    m.test_waveform_chunk_parallel_random_sequences(omx, oracle, seed)
