"""HIP drop-in plugins for the records-backed hot path.

``hip_default()`` is the packaging the reference uses for backends: a profile function
returning plugin instances (waveform_analysis/core/plugins/profiles.py:20-62); register them
with ``ctx.register(p, allow_override=True)`` to replace the CPU plugins of the same name.
"""

from .basic_features import HipBasicFeaturesPlugin
from .filtered_waveforms import HipFilteredWaveformsPlugin
from .hit_finder import HipHitFinderPlugin
from .hit_grouped import HipHitGroupedPlugin
from .hit_merge import HipHitMergeClustersPlugin, HipHitMergedComponentsPlugin, HipHitMergePlugin
from .s1_s2 import HipS1S2ClassifierPlugin
from .signal_peaks import HipSignalPeaksStreamPlugin
from .threshold_hit import HipThresholdHitPlugin
from .wave_pool_filtered import HipWavePoolFilteredPlugin
from .waveform_width import HipWaveformWidthPlugin
from .width_integral import HipWaveformWidthIntegralPlugin


def hip_default():
    return [HipWavePoolFilteredPlugin(), HipThresholdHitPlugin(), HipBasicFeaturesPlugin(),
            HipWaveformWidthIntegralPlugin(), HipHitGroupedPlugin(), HipHitFinderPlugin(),
            HipFilteredWaveformsPlugin(), HipWaveformWidthPlugin(), HipS1S2ClassifierPlugin(),
            HipHitMergeClustersPlugin(), HipHitMergePlugin(), HipHitMergedComponentsPlugin(),
            HipSignalPeaksStreamPlugin()]


__all__ = ["HipWavePoolFilteredPlugin", "HipThresholdHitPlugin", "HipBasicFeaturesPlugin",
           "HipWaveformWidthIntegralPlugin", "HipHitGroupedPlugin", "HipHitFinderPlugin", "HipFilteredWaveformsPlugin",
           "HipWaveformWidthPlugin", "HipS1S2ClassifierPlugin", "HipHitMergeClustersPlugin",
           "HipHitMergePlugin", "HipHitMergedComponentsPlugin", "HipSignalPeaksStreamPlugin", "hip_default"]
