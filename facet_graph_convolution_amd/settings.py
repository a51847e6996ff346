"""Constants of the reference's ``settings.py`` that are parity inputs of the hot path."""
K_faces = 23                # settings.py:23
COARSENING_STEPS = 2        # settings.py:31
COARSENING_LVLS = 3         # settings.py:32
MAX_PATCH_SIZE = 20000      # settings.py:20 (reference only; meshes are kept whole here)
MIN_PATCH_SIZE = 2000       # settings.py:22
SAVEITER = 5000             # settings.py:30
NUM_ITERATIONS = 300000     # settings.py:33
MAX_EDGES = 20              # dataClasses.py:40 (getEdgeMap(faces0, maxEdges = 20)); train.py:44 reads it off v_e_map


def getGTFilename(filename):
    """settings.py:44-47: the ground truth of `<model>_n<k>.obj` is `<model>.obj` (the last 7 characters go)."""
    return filename[:-7] + ".obj"
