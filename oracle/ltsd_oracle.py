"""CPU restatement (numpy float64) of the LTSD measure behind the reference's voice-activity front
end.  TEST INFRASTRUCTURE ONLY -- never imported by the product.

PARITY UNPINNED.  The reference (src/filters/ltsd.py:14,36-38,74-75) calls third-party
``pyssp.vad.ltsd.LTSD`` -- not vendored, not installed, version unpinned, and its
``compute_with_noise`` returning (intervals, ltsds) exists only in the authors' fork.  What follows
restates the algorithm as published (Ramirez et al., "Efficient voice activity detection algorithms
using long-term speech information", 2004) in the form pyssp gives it:

    windows  = len(signal) / (winsize/2) - 1                     (integer division)
    frame_l  = signal[l*winsize/2 : l*winsize/2 + winsize]
    amp_l    = |FFT_winsize(frame_l * window)|                   (exact length, all winsize bins)
    noise    = mean_l amp_l(noise signal)
    LTSE_l   = max_{|j|<=order} amp_{l+j}
    LTSD_l   = 10 log10( sum_k LTSE_l[k]^2 / noise[k]^2 / winsize ),   0 if l < order or l + order >= windows

and the call-site constants of the reference: winsize = int(0.04644 fs), window = numpy.hanning,
order 5, lambda0 = 1.1 max LTSD(noise), lambda1 = 2 lambda0 (ltsd.py:17-21,39-41,66-69).  Frames
that would run past the end of the signal are zero-extended."""
import numpy as np

MAGIC_NUMBER = 0.04644


def window_size(fs):
    return int(MAGIC_NUMBER * fs)


def num_windows(n, winsize):
    return max(0, n // (winsize // 2) - 1)


def amplitudes(signal, winsize):
    signal = np.asarray(signal, dtype=np.float64)
    win = np.hanning(winsize)
    shift = winsize // 2
    wn = num_windows(len(signal), winsize)
    out = np.zeros((wn, winsize))
    for l in range(wn):
        fr = signal[l * shift:l * shift + winsize]
        if len(fr) < winsize:
            fr = np.concatenate([fr, np.zeros(winsize - len(fr))])
        out[l] = np.abs(np.fft.fft(fr * win))
    return out


def noise_spectrum(noise, winsize):
    return amplitudes(noise, winsize).mean(axis=0)


def ltsd(signal, noise_amp, winsize, order=5):
    amp = amplitudes(signal, winsize)
    wn = amp.shape[0]
    out = np.zeros(wn)
    for l in range(wn):
        if l < order or l + order >= wn:
            continue
        ltse = amp[l - order:l + order + 1].max(axis=0)
        out[l] = 10.0 * np.log10(np.sum(ltse ** 2 / noise_amp ** 2) / float(winsize))
    return out


def thresholds(noise, winsize, order=5):
    na = noise_spectrum(noise, winsize)
    l = ltsd(noise, na, winsize, order)
    lam0 = 1.1 * (l.max() if len(l) else 0.0)
    return na, lam0, 2.0 * lam0
