"""Shim package: see trace_amd/compat/README.md."""
