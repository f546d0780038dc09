"""Host-side mirror of the reference's `environments` package for the env step path:
same module names, class names, config keys and call signatures; every number is
computed by the HIP kernels behind include/qd.h."""
