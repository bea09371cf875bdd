"""DataCollection — train/test/validation split, host mirror of S/DataCollection.scala."""
import numpy as np

from .dataset import DataSet


class DataCollection:
    """S/DataCollection.scala:9-25."""

    def __init__(self, trainingSet, testSet, validationSet, numFeature=0):
        self.trainingSet, self.testSet, self.validationSet = trainingSet, testSet, validationSet
        self.numFeature = numFeature

    @property
    def dimension(self):  # :15-23
        if self.numFeature > 0:
            return self.numFeature
        return max(self.trainingSet.dimension, self.testSet.dimension, self.validationSet.dimension)

    @staticmethod
    def splitByRandom(rawData, trainWeight, testWeight, validateWeight=0.0, seed=0, **dataset_kw):
        """S/DataCollection.scala:29-51: RDD.randomSplit(weights) — every row is sent to one split by an
        independent uniform draw against the normalised cumulative weights.  The reference passes
        DataSet.dimension(rawData) as numFeature, which returns the ROW COUNT (quirk Q8); that is not
        reproduced: numFeature stays 0 and `dimension` is the largest feature index of the splits."""
        if trainWeight == 0 or testWeight == 0:
            raise Exception("Both TrainingSet and TestSet are required")   # :35-37
        weights = [trainWeight, testWeight] + ([validateWeight] if validateWeight > 0 else [])
        edges = np.cumsum(weights) / float(np.sum(weights))
        rng = np.random.Generator(np.random.PCG64(seed))
        which = np.searchsorted(edges, rng.random(rawData.size), side="right")
        which = np.minimum(which, len(weights) - 1)
        parts = []
        for s, name in enumerate(("TrainingSet", "TestSet", "ValidationSet")):
            rows = np.nonzero(which == s)[0] if s < len(weights) else np.zeros(0, np.int64)
            # only the training split is trained on: the others are scored (fm.computeRMSE(testSet),
            # S/driver.scala:111) and keep just their rows and labels on the device
            kw = dict(dataset_kw) if s == 0 else dict(dict(dataset_kw, batch_rows=0), scoring=True)
            parts.append(_take_rows(rawData, rows, name, kw))
        return DataCollection(parts[0], parts[1], parts[2], 0)


def _take_rows(ds, rows, name, kw):
    lens = ds.row_ptr[rows + 1] - ds.row_ptr[rows]
    ptr = np.zeros(len(rows) + 1, np.int64)
    np.cumsum(lens, out=ptr[1:])
    # entry indices of the kept rows, row by row: start of the row + position inside it
    idx = np.repeat(ds.row_ptr[rows], lens) + (np.arange(int(ptr[-1]), dtype=np.int64) - np.repeat(ptr[:-1], lens))
    return DataSet(ptr, ds.col[idx], ds.val[idx], ds.y[rows], name=name, **kw)
