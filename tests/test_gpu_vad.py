"""LTSD voice-activity front end ("next" row f-3) through the C ABI vs the numpy restatement of
the published algorithm (oracle/ltsd_oracle.py -- parity unpinned: the reference's own
implementation is a third-party package that is not in its tree)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _scene(fs, seed, bursts):
    """Stationary noise with speech bursts at the given (start_s, end_s) intervals."""
    from speaker_recognition_amd import synth
    rng = np.random.default_rng(seed)
    total = int(max(e for _, e in bursts) * fs + 1.5 * fs)
    sig = rng.normal(0.0, 120.0, total)
    for i, (s, e) in enumerate(bursts):
        sp = synth.synth_speech(3 + i, e - s, fs, seed=50 + i).astype(np.float64)
        sig[int(s * fs):int(s * fs) + len(sp)] += sp
    noise = rng.normal(0.0, 120.0, 3 * fs)
    return np.clip(sig, -32768, 32767).astype(np.int16), np.clip(noise, -32768, 32767).astype(np.int16)


@pytest.mark.parametrize("fs", [8000, 16000])
def test_ltsd_values_vs_oracle(built_lib, fs):
    """Window size int(0.04644 fs) = 371 / 743 (odd, 743 prime): the exact-length DFT, the long-term
    envelope and the dB measure against the float64 restatement; ragged signals incl. ones too
    short for a single window."""
    from oracle import ltsd_oracle as lo
    from speaker_recognition_amd.filters import ltsd as L
    sig, noise = _scene(fs, 1, [(1.0, 2.2), (3.0, 3.6)])
    N = lo.window_size(fs)
    na_want = lo.noise_spectrum(noise, N)
    na = L.noise_spectrum(noise, N)
    assert na.shape == (N // 2 + 1,)
    assert np.max(np.abs(na - na_want[:N // 2 + 1]) / na_want[:N // 2 + 1]) < 1e-4
    sigs = [sig, sig[:fs // 3], sig[5000:5000 + N + N // 2], np.zeros(10, np.int16), sig[::-1].copy()]
    got = L.ltsd_values(sigs, na_want[:N // 2 + 1].astype(np.float32), N, 5)
    for s, g in zip(sigs, got):
        want = lo.ltsd(s, na_want, N, 5)
        assert g.shape == want.shape == (lo.num_windows(len(s), N),)
        if len(want):
            assert np.max(np.abs(g - want)) < 2e-3, np.max(np.abs(g - want))      # dB
    assert np.max(got[0]) > 20.0 and np.all(got[0][:5] == 0) and np.all(got[0][-5:] == 0)


def test_vad_filter_finds_the_bursts(built_lib):
    """VAD.init_noise / filter (filters/VAD.py surface): the intervals cover the speech bursts and
    little else; ModelInterface.filter applies the reference's one-third rule."""
    from speaker_recognition_amd.filters import VAD
    from speaker_recognition_amd.interface import ModelInterface
    fs = 16000
    bursts = [(1.0, 2.5), (4.0, 5.0)]
    sig, noise = _scene(fs, 2, bursts)
    v = VAD()
    with pytest.raises(RuntimeError):
        v.filter(fs, sig)
    v.init_noise(fs, noise)
    assert v.ltsd.lambda1 == 2 * v.ltsd.lambda0 > 0
    voiced, intervals = v.filter(fs, sig)
    mask = np.zeros(len(sig), bool)
    for s, e in intervals:
        mask[s:e] = True
    truth = np.zeros(len(sig), bool)
    for s, e in bursts:
        truth[int(s * fs):int(e * fs)] = True
    assert len(voiced) == int(mask.sum())
    assert (mask & truth).sum() > 0.85 * truth.sum()              # the bursts are found
    assert (mask & ~truth).sum() < 0.25 * truth.sum()             # hang-over of +-order windows only
    nv, ni = v.filter(fs, noise[:2 * fs])
    assert len(nv) == 0 and ni == []                              # noise alone: nothing voiced
    stereo = np.stack([sig, np.zeros_like(sig)], axis=1)
    v2, i2 = v.filter(fs, stereo)                                 # first channel (ltsd.py:79-82)
    assert i2 == intervals
    m = ModelInterface(verbose=False)
    with pytest.raises(RuntimeError):
        m.filter(fs, sig)
    m.init_noise(fs, noise)
    kept = m.filter(fs, sig)
    assert len(kept) == len(voiced)                               # > 1/3 of the signal is voiced
    assert len(m.filter(fs, np.concatenate([noise, noise, sig[:int(2.6 * fs)]]))) == 0    # < 1/3: dropped
    many = v.ltsd.filter_many([sig, noise[:fs], sig[:3 * fs]])
    assert many[0][1] == intervals and many[1][1] == []
