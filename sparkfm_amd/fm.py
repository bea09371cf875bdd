"""FM façade + fit loop — host mirror of S/fm/FM.scala and S/fm/impl/FactorizationMachines.scala."""
import logging

from .model import FMModel

log = logging.getLogger("sparkfm_amd")


class Task:
    """S/Task.scala:3-6 (accepted and, as in the reference, never read)."""
    Regression = "Regression"
    Classification = "Classification"


class FactorizationMachines:
    """S/fm/impl/FactorizationMachines.scala:9-53."""

    def __init__(self, dataset, numFactor=8, task=Task.Regression, maxIteration=100, timeout=0, seed=0):
        self.dataset = dataset
        self.numFactor = numFactor
        self.task = task
        self.maxIteration = maxIteration
        self.timeout = timeout
        self.seed = seed
        self.rmse_history = []

    def withRelation(self, relation):
        # S/fm/bs/* is an unfinished stub in the reference (Relation.cache = null); out of scope.
        raise NotImplementedError("relational (block-structure) FM is a non-functional stub in SparkFM")

    def learnWith(self, fml, init=None):
        """S/fm/impl/FactorizationMachines.scala:30-51: cache; new FMModel(dimension, numFactor);
        maxIteration x { computeRMSE (logged); fm = fml.learn(fm, dataset) }; unpersist.
        `init` = (w0, w, v) injects parameters (the reference's own init is unseeded, quirk Q2)."""
        ds = self.dataset.cache()                                     # :36
        fm = FMModel(ds.dimension, self.numFactor, seed=self.seed, device=ds.device)   # :39
        if init is not None:
            fm.w0, fm.w, fm.v = init
        for i in range(1, self.maxIteration + 1):                     # :42
            rmse = fm.computeRMSE(ds)                                 # :43
            self.rmse_history.append(rmse)
            log.info("Iteration %d in progress... (%s RMSE = %.6f)", i, ds.name, rmse)
            fm = fml.learn(fm, ds)                                    # :45
        ds.unpersist()                                                # :48
        return fm


def FM(dataset, numFactor, task=Task.Regression, maxIteration=100, timeout=0, seed=0):
    """FM.apply (S/fm/FM.scala:25-33)."""
    return FactorizationMachines(dataset, numFactor, task, maxIteration, timeout, seed)
