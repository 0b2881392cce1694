#!/usr/bin/env python3
"""oracle/make_ref_py.py -- TEST INFRASTRUCTURE ONLY.  Build recipe for oracle/_ref/ (git-ignored, like
the reference DSO built by oracle/Makefile): turns the reference's OWN Python binding
(/root/reference/src/gmm/python/pygmm.py) and its speaker-set classes
(/root/reference/src/testbench/gmmset.py) into importable Python-3 modules bound to a library of our
choice, by mechanical edits only -- nothing is rewritten, nothing lands in the repository's history:

  pygmm.py   * the library path (pygmm.py:16) comes from the environment variable SR_REF_BINDING_LIB
             * the restype stub of INTEGRATION.md section 1 after line 31, and c_void_p(...) around the
               two stored handles (the 64-bit handle truncation the reference has on any modern Python)
             * Python 2 -> 3: the print statement (:113), c_char_p(str) -> c_char_p(str.encode())
               (:61, :66), the 'wb' temp file of loads() (:79)
  gmmset.py  * `from gmm.python.pygmm import GMM` -> the module generated above
             * Python 2 -> 3: iteritems -> items, map(...) -> list(map(...))

tests/test_gpu_reference_binding.py imports the results and runs the reference's own fit / score /
score_all / GMMSetPyGMM.predict_one / before_pickle code against lib/pygmm.so on the GPU.
Usage: python oracle/make_ref_py.py   (only where /root/reference exists)"""
import os
import sys

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")


def edit(src, old, new, count):
    assert src.count(old) == count, (old, src.count(old))
    return src.replace(old, new)


def main():
    if not os.path.isdir(REF):
        print("make_ref_py: %s not present, nothing generated" % REF)
        return 0
    os.makedirs(OUT, exist_ok=True)
    s = open(os.path.join(REF, "src/gmm/python/pygmm.py")).read()
    s = edit(s, "pygmm = cdll.LoadLibrary(path.join(dirname, '../lib/pygmm.so'))",
             "pygmm = cdll.LoadLibrary(os.environ['SR_REF_BINDING_LIB'])", 1)
    s = edit(s, "pygmm.score_instance.restype = c_double\n",
             "pygmm.score_instance.restype = c_double\n"
             "pygmm.new_gmm.restype = c_void_p          # INTEGRATION.md section 1\n"
             "pygmm.load.restype = c_void_p\n"
             "pygmm.new_gmm.argtypes = [c_int, c_int]\n"
             "pygmm.load.argtypes = [c_char_p]\n", 1)
    s = edit(s, "self.gmm = pygmm.new_gmm(c_int(nr_mixture), c_int(covariance_type))",
             "self.gmm = c_void_p(pygmm.new_gmm(nr_mixture, covariance_type))", 1)
    s = edit(s, "gmm.gmm = pygmm.load(c_char_p(model_file))", "gmm.gmm = c_void_p(pygmm.load(model_file.encode()))", 1)
    s = edit(s, "pygmm.dump(self.gmm, c_char_p(model_file))", "pygmm.dump(self.gmm, c_char_p(model_file.encode()))", 1)
    s = edit(s, "print 'training from ubm ...'", "print('training from ubm ...')", 1)
    s = edit(s, "f = open(tmp_file, 'wb')", "f = open(tmp_file, 'w')", 1)
    open(os.path.join(OUT, "ref_pygmm_py3.py"), "w").write(s)
    g = open(os.path.join(REF, "src/testbench/gmmset.py")).read()
    g = edit(g, "from gmm.python.pygmm import GMM", "from ref_pygmm_py3 import GMM", 1)
    g = edit(g, ".iteritems()", ".items()", 2)
    g = edit(g, "return map(self.predict_one, X)", "return list(map(self.predict_one, X))", 1)
    g = edit(g, "scores = map(lambda v: v / x_len, scores)", "scores = list(map(lambda v: v / x_len, scores))", 1)
    g = edit(g, "return map(self.predict_one_with_rejection, X)", "return list(map(self.predict_one_with_rejection, X))", 1)
    open(os.path.join(OUT, "ref_gmmset_py3.py"), "w").write(g)
    print("make_ref_py: wrote", os.listdir(OUT))
    return 0


if __name__ == "__main__":
    sys.exit(main())
