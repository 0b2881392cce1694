"""Command line -- mirrors the reference's ``src/speaker-recognition.py`` (:21-100):

    speaker-recognition.py -t enroll  -i "./bob/ ./mary/ ./person*" -m model.out
    speaker-recognition.py -t predict -i "./*.wav" -m model.out

Wav files in each input directory are labelled with the directory's basename; wildcard inputs
must be quoted (they go to glob).  Extra options (not in the reference) select the feature
framing / model order so that the BASELINE.json configs can be run from the shell.
"""
from __future__ import annotations

import argparse
import glob
import itertools
import os
import sys

from scipy.io import wavfile

from .interface import ModelInterface


def read_wav(fname):
    """src/gui/utils.py:10-13."""
    fs, signal = wavfile.read(fname)
    assert len(signal.shape) == 1, "Only Support Mono Wav File!"
    return fs, signal


def get_args(argv=None):
    parser = argparse.ArgumentParser(description="Speaker Recognition Command Line Tool",
                                     formatter_class=argparse.RawDescriptionHelpFormatter)
    parser.add_argument("-t", "--task", help='Task to do. Either "enroll" or "predict"', required=True)
    parser.add_argument("-i", "--input", help="Input Files(to predict) or Directories(to enroll)", required=True)
    parser.add_argument("-m", "--model", help="Model file to save(in enroll) or use(in predict)", required=True)
    parser.add_argument("--mixtures", type=int, default=32, help="GMM order per speaker (default 32, gmmset.py:16)")
    parser.add_argument("--win-length-ms", type=float, default=32)
    parser.add_argument("--win-shift-ms", type=float, default=16)
    parser.add_argument("--fft-size", type=int, default=2048)
    parser.add_argument("--deltas", type=int, default=0, choices=[0, 1, 2], help="append delta orders")
    parser.add_argument("--no-lpc", action="store_true", help="MFCC half of mix_feature only (13 dims)")
    parser.add_argument("--seed", type=int, default=-1, help="EM initialisation seed (-1: random)")
    parser.add_argument("--device", type=int, default=0)
    parser.add_argument("--gpus", type=int, default=1,
                        help="predict: shard the input files over this many GPUs from one process (0 = all visible)")
    return parser.parse_args(argv)


def task_enroll(input_dirs, output_model, args=None):
    m = _make_interface(args)
    input_dirs = [os.path.expanduser(k) for k in input_dirs.strip().split()]
    dirs = itertools.chain(*(glob.glob(d) for d in input_dirs))
    dirs = [d for d in dirs if os.path.isdir(d)]
    if len(dirs) == 0:
        print("No valid directory found!")
        sys.exit(1)
    training_stats = []
    for d in dirs:
        label = os.path.basename(d.rstrip("/"))
        wavs = sorted(glob.glob(d + "/*.wav"))
        if len(wavs) == 0:
            print("No wav file found in {0}".format(d))
            continue
        print("Label '{0}' has files: {1}".format(label, ", ".join(wavs)))
        total_len = 0
        for wav in wavs:
            fs, signal = read_wav(wav)
            print("   File '{}' has frequency={} and length={}".format(wav, fs, len(signal)))
            total_len += len(signal)
            m.enroll(label, fs, signal)
        training_stats.append((label, total_len))
    print("--------------------------------------------")
    for label, total_len in training_stats:
        print("Total length of training data for '{}' is {}".format(label, total_len))
    print("For best accuracy, please make sure all labels have similar amount of training data!")
    m.train()
    m.dump(output_model)


def task_predict(input_files, input_model, gpus=1):
    m = ModelInterface.load(input_model)
    out = []
    files = sorted(glob.glob(os.path.expanduser(input_files)))
    if gpus != 1:
        # every file in one utterance-sharded pass over the node's GPUs (interface.predict_many)
        labels = m.predict_many([read_wav(f) for f in files], gpus=gpus)
        for f, label in zip(files, labels):
            print(f, "->", label)
            out.append((f, label))
        return out
    for f in files:
        fs, signal = read_wav(f)
        label = m.predict(fs, signal)
        print(f, "->", label)
        out.append((f, label))
    return out


def _make_interface(args):
    if args is None:
        return ModelInterface()
    fk = {}
    if args.win_length_ms != 32:
        fk["win_length_ms"] = args.win_length_ms
    if args.win_shift_ms != 16:
        fk["win_shift_ms"] = args.win_shift_ms
    if args.fft_size != 2048:
        fk["FFT_SIZE"] = args.fft_size
    return ModelInterface(gmm_order=args.mixtures, feature_kwargs=fk, diff=args.deltas > 0,
                          nd=max(1, args.deltas), lpc=not args.no_lpc and args.deltas == 0,
                          gmm_kwargs={"seed": args.seed})


def main(argv=None):
    args = get_args(argv)
    from . import _lib
    _lib.set_device(args.device)
    if args.task == "enroll":
        task_enroll(args.input, args.model, args)
    elif args.task == "predict":
        task_predict(args.input, args.model, args.gpus)
    else:
        print('task must be "enroll" or "predict"')
        sys.exit(2)


if __name__ == "__main__":
    main()
