"""Common estimator base: scikit-learn parameter handling plus the ``summarize()`` hook
that MSMBuilder's CLI prints after ``fit`` (reference: msmbuilder/base.py)."""
import sklearn.base


class BaseEstimator(sklearn.base.BaseEstimator):
    def summarize(self):
        """One block of human-readable diagnostics; subclasses override."""
        return 'NotImplemented'
