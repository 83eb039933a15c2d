"""Host-side feature preparation (CPU string work, outside the hot path): a dependency-free
restatement of what inference needs from the reference's CriteoDataPreprocessor and of its
synthetic data generator, so that BASELINE config 0 ("--use_synthetic --n_samples 10000") can be
reproduced without pandas / sklearn pickles.

* synthetic_criteo   data_preprocessing.py:242-289  np.random.seed(42); lognormal numericals;
                     categorical strings 'cat_<j>' with cardinalities [1000,500,100,50]*6+[20,10]
* fit_preprocessor   data_preprocessing.py:88-142   median fill -> log1p(|x|) -> StandardScaler;
                     categories seen < 10 times -> 'rare'; LabelEncoder (sorted unique classes)
"""
from __future__ import annotations

from typing import Dict

import numpy as np

from .pipeline import Preprocessor
from .synth import CRITEO_SYNTH_CARDS, NUM_COLS

CAT_COLS = [f"C{i}" for i in range(1, 27)]


def synthetic_criteo(n_samples: int = 10000, seed: int = 42):
    """-> (numerical float64 [n,13], categorical {col: array of str}, labels int [n])."""
    rng = np.random.RandomState(seed)                                     # np.random.seed(42) (:250)
    numerical = rng.lognormal(0, 1, size=(n_samples, 13))                 # :253
    categorical = {}
    for i, col in enumerate(CAT_COLS):                                    # :262-266
        names = np.array([f"cat_{j}" for j in range(CRITEO_SYNTH_CARDS[i])])
        categorical[col] = rng.choice(names, size=n_samples)
    feature_sum = numerical[:, 0] + numerical[:, 1]                       # :270-272
    probs = 1 / (1 + np.exp(-0.1 * (feature_sum - 5)))
    labels = (rng.random_sample(n_samples) < probs).astype(int)
    return numerical, categorical, labels


def fit_preprocessor(numerical: np.ndarray, categorical: Dict[str, np.ndarray], min_count: int = 10):
    """-> (Preprocessor, numerical_scaled float32 [n,13], categorical_encoded int64 [n,26])."""
    num = np.array(numerical, dtype=np.float64)
    med = np.nanmedian(num, axis=0)
    num = np.where(np.isnan(num), med, num)
    num = np.log1p(np.abs(num))
    mean, scale = num.mean(axis=0), num.std(axis=0)
    scale = np.where(scale > 0, scale, 1.0)
    classes, enc = {}, []
    for col in CAT_COLS:
        v = np.asarray(categorical[col]).astype(str)
        uniq, cnt = np.unique(v, return_counts=True)
        rare = set(uniq[cnt < min_count].tolist())
        if rare:
            v = np.where(np.isin(v, list(rare)), "rare", v)
        cls = np.unique(v)                                                # LabelEncoder: sorted classes
        classes[col] = cls.tolist()
        enc.append(np.searchsorted(cls, v))
    pp = Preprocessor(classes, list(NUM_COLS), mean, scale)
    return pp, ((num - mean) / scale).astype(np.float32), np.stack(enc, axis=1).astype(np.int64)
