"""Process-wide switches of runia_core_amd (all default to the reference's behaviour)."""

# Fit covariance / precision matrices of setup() on the GPU (see runia_core_amd.device_fit).
device_fit = False
