"""Stand-alone mirror of the vLLM interfaces the PQ linear path plugs into, used when vLLM itself is
not importable (this build's tests and benchmarks) -- same class names, argument meaning and error
behaviour as the reference, so that code written against one runs against the other.
"""
