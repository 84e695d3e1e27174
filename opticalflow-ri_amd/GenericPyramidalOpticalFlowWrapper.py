"""Class form of the pyramid driver: the constructor keeps the settings, ``calculateFlow(im1, im2)`` runs
``genericPyramidalOpticalFlow`` with them (interface of the reference's src/GenericPyramidalOpticalFlowWrapper.py:8-64,
which its benchmark script uses, benchmark_of_methods.py:166-175).  ``calculateFlowDevice`` is the same on frames that
are already in GPU memory."""
from GenericPyramidalOpticalFlow import genericPyramidalOpticalFlow, pyramidal_flow_device

# constructor keyword -> keyword of the driver
_TO_DRIVER = {
    "pyr_levels": "pyramidalLevels",
    "k_levels": "kLevels",
    "filter_opt": "FILTER_OPT",
    "optional_algo_adapter": "optionalOFlowAlgoAdapter",
    "warping": "warping",
    "bi_linear": "biLinear",
    "pyramidal_intermediate_scaling": "pyramidalIntermediateScaling",
    "pyramidal_scaling": "pyramidalScaling",
}


class GenericPyramidalOpticalFlowWrapper(object):
    def __init__(self, algo_adapter, filter_sigma=0.0, pyr_levels=1, k_levels=1, filter_opt=None,
                 optional_algo_adapter=None, warping=True, bi_linear=True, pyramidal_intermediate_scaling=True,
                 pyramidal_scaling=False):
        given = dict(locals())
        given.pop("self")
        vars(self).update(given)         # attributes carry the constructor's names, as on the reference's object

    def _driver_args(self):
        return {theirs: getattr(self, ours) for ours, theirs in _TO_DRIVER.items()}

    def calculateFlow(self, im1, im2):
        """(U, V) as NumPy arrays; adapters that consume parameter lists are consumed by every call."""
        return genericPyramidalOpticalFlow(im1, im2, self.filter_sigma, self.algo_adapter, **self._driver_args())

    def calculateFlowDevice(self, im1, im2):
        """The same on DeviceArrays, returning DeviceArrays without waiting for the GPU."""
        return pyramidal_flow_device(im1, im2, self.filter_sigma, self.algo_adapter, **self._driver_args())
