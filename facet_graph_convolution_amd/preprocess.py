"""Build the pickled training / validation sets from folders of OBJ files (the reference's preprocess.py:8-52):

    python -m facet_graph_convolution_amd.preprocess TRAINING_DIR GT_DIR DUMP_DIR [--valid VALID_DIR] [--redundancy R]

Every `name.obj` of TRAINING_DIR is paired with its ground truth through `gt_name` (default: the reference's
`getGTFilename`, settings.py:44-47, `<model>_n<k>.obj` -> `<model>.obj`; `gt_filename` below is a laxer variant), preprocessed
natively (adjacency, coarsening, padding) `redundancy` times (each pass draws a different coarsening: the reference
uses this as data augmentation, settings.py:24) and pickled as `trainingSet.pkl` / `validSet.pkl`.
"""
import argparse
import os
import pickle
import re

from .dataClasses import TrainingSet
from .settings import getGTFilename


def gt_filename(noisy_name):
    """`bunny_n1.obj`, `bunny_noisy.obj`, `bunny_n.obj` -> `bunny.obj`; names without such a suffix map to themselves."""
    stem = noisy_name[:-4]
    stem = re.sub(r"(_n\d*|_noisy\d*)$", "", stem)
    return stem + ".obj"


def pickleData(training_dir, gt_dir, dump_dir, valid_dir=None, redundancy=1, gt_name=getGTFilename, log=print):
    os.makedirs(dump_dir, exist_ok=True)
    out = {}
    for tag, folder, rep in (("trainingSet.pkl", training_dir, redundancy), ("validSet.pkl", valid_dir, 1)):
        if not folder or not os.path.isdir(folder):
            continue
        ds = TrainingSet()
        for f in sorted(os.listdir(folder)):
            if not f.endswith(".obj"):
                continue
            log("Adding %s (%i)" % (f, ds.mesh_count))
            for _ in range(rep):
                ds.addMeshWithGT(folder, f, gt_dir, gt_name(f))
        if ds.mesh_count:
            with open(os.path.join(dump_dir, tag), "wb") as fp:
                pickle.dump(ds, fp)
            out[tag] = ds
    return out


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("training_dir")
    ap.add_argument("gt_dir")
    ap.add_argument("dump_dir")
    ap.add_argument("--valid", default=None)
    ap.add_argument("--redundancy", type=int, default=1)
    args = ap.parse_args(argv)
    pickleData(args.training_dir, args.gt_dir, args.dump_dir, args.valid, args.redundancy)
    print("Preprocessing complete. Dump files saved to " + args.dump_dir)


if __name__ == "__main__":
    main()
