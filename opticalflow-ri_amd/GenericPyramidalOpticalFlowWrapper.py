"""Class form of the pyramid driver (reference src/GenericPyramidalOpticalFlowWrapper.py:8-64)."""
from GenericPyramidalOpticalFlow import genericPyramidalOpticalFlow


class GenericPyramidalOpticalFlowWrapper:
    def __init__(self, algo_adapter, filter_sigma=0.0, pyr_levels=1, k_levels=1, filter_opt=None,
                 optional_algo_adapter=None, warping=True, bi_linear=True, pyramidal_intermediate_scaling=True,
                 pyramidal_scaling=False):
        self.algo_adapter = algo_adapter
        self.filter_sigma = filter_sigma
        self.pyr_levels = pyr_levels
        self.k_levels = k_levels
        self.filter_opt = filter_opt
        self.optional_algo_adapter = optional_algo_adapter
        self.warping = warping
        self.bi_linear = bi_linear
        self.pyramidal_intermediate_scaling = pyramidal_intermediate_scaling
        self.pyramidal_scaling = pyramidal_scaling

    def calculateFlow(self, im1, im2):
        return genericPyramidalOpticalFlow(im1, im2, self.filter_sigma, self.algo_adapter,
                                           pyramidalLevels=self.pyr_levels, kLevels=self.k_levels,
                                           FILTER_OPT=self.filter_opt,
                                           optionalOFlowAlgoAdapter=self.optional_algo_adapter,
                                           warping=self.warping, biLinear=self.bi_linear,
                                           pyramidalIntermediateScaling=self.pyramidal_intermediate_scaling,
                                           pyramidalScaling=self.pyramidal_scaling)
