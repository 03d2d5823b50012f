"""The callers of the importance-sampling estimator in the reference's evaluate.py, as functions of a model object.

Only what sits directly on the path: the 2-D grid evaluation (evaluate.py:15-33, the shape of BASELINE config 2) and
the outlier-detection score (evaluate.py:35-59, Roundtrip's own part: the One-class-SVM / Isolation-Forest baselines
and all plotting are out of scope).  `model` is a roundtrip_amd.RoundtripModel (or anything with its
estimate_py_with_IS signature)."""
import numpy as np
from scipy.stats import rankdata

# (range, sd_y, scale) per simulated data set: evaluate.py:234-239 with find_y_sampler's constants (:134-141)
GRID_SETTINGS = {
    "indep_gmm": ((-1.5, 1.5, -1.5, 1.5), 0.05, 0.5),
    "eight_octagon_gmm": ((-5, 5, -5, 5), 0.1, 0.5),
    "involute": ((-6, 5, -5, 5), 0.4, 0.5),
}


def create_2d_grid_data(x1_min, x1_max, x2_min, x2_max, n=100):
    # evaluate.py:15-20
    grid_x1 = np.linspace(x1_min, x1_max, n)
    grid_x2 = np.linspace(x2_min, x2_max, n)
    v1, v2 = np.meshgrid(grid_x1, grid_x2)
    data_grid = np.vstack((v1.ravel(), v2.ravel())).T
    return v1, v2, data_grid


def visualization_2d(model, x1_min, x1_max, x2_min, x2_max, sd_y, scale, n=100, epoch=0, sample_size=10000,
                     save_path=None):
    """evaluate.py:23-33 without the imshow: the n x n density p(y) on the grid (log=False, the default --ss of 10000,
    evaluate.py:225).  With save_path the array goes to '<save_path>/2d_grid_density_pre.npy' (the reference writes
    the .png of the same stem)."""
    v1, v2, data_grid = create_2d_grid_data(x1_min, x1_max, x2_min, x2_max, n)
    py = model.estimate_py_with_IS(data_grid, epoch, sd_y=sd_y, scale=scale, sample_size=sample_size, log=False,
                                   save=False)
    py = py.reshape((n, n))
    if save_path is not None:
        np.save('%s/2d_grid_density_pre.npy' % save_path.rstrip('/'), py)
    return v1, v2, py


def precision_at_K(score, label_test):
    # evaluate.py:55-59: among the K lowest scores (K = number of outliers), the fraction that are outliers
    rank = rankdata(score)
    nb_test = np.sum(label_test)
    precision = len([1 for item in zip(rank, label_test) if item[0] <= nb_test and item[1] == 1]) * 1.0 / nb_test
    return precision


def odd_evaluate(model, X_test, label_test, best_sd, best_scale, epoch=0, sample_size=40000):
    """evaluate.py:49-51: log p(y) of the test points as the outlier score, precision at K."""
    py = model.estimate_py_with_IS(X_test, epoch, sd_y=best_sd, scale=best_scale, sample_size=sample_size, log=True,
                                   save=False)
    return precision_at_K(py, label_test), py
