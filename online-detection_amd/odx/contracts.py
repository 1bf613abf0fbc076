"""The reference's abstract module APIs (ClassifierAbstract, RegionClassifierAbstract,
RegionRefinerAbstract, FeatureExtractorAbstract) as generated ABCs: a name plus the list of
methods a concrete module must provide (AccuracyEvaluatorAbstract likewise)."""
from abc import ABC, abstractmethod


def make_contract(name, methods, doc):
    ns = {"__doc__": doc, "__init__": lambda self: None}
    for m in methods:
        def stub(self, *args, _m=m, **kwargs):
            raise NotImplementedError(_m)
        stub.__name__ = m
        ns[m] = abstractmethod(stub)
    return type(name, (ABC,), ns)


ClassifierAbstract = make_contract(
    "ClassifierAbstract", ["train", "predict", "test"],
    "classifier plug-in of OnlineRegionClassifier (reference: region-classifier/ClassifierAbstract.py:4-18)")
RegionClassifierAbstract = make_contract(
    "RegionClassifierAbstract", ["loadRegionClassifier", "trainRegionClassifier", "testRegionClassifier", "predict"],
    "region classifier module (reference: src/modules/RegionClassifierAbstract.py:9-41)")
RegionRefinerAbstract = make_contract(
    "RegionRefinerAbstract", ["loadRegionRefiner", "trainRegionRefiner", "testRegionRefiner", "predict"],
    "region refiner module (reference: src/modules/RegionRefinerAbstract.py:4-22)")
FeatureExtractorAbstract = make_contract(
    "FeatureExtractorAbstract", ["extractFeatures"],
    "feature extractor module (reference: src/modules/FeatureExtractorAbstract.py:4-22)")
AccuracyEvaluatorAbstract = make_contract(
    "AccuracyEvaluatorAbstract", ["evaluateAccuracyDetection"],
    "accuracy evaluator module (reference: src/modules/accuracy-evaluator/AccuracyEvaluatorAbstract.py:4-10)")
