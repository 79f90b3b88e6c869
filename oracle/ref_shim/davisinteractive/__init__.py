"""Test-infrastructure stand-in for the `davisinteractive` package (the DAVIS interactive-track evaluation framework, an
un-vendored third-party dependency of the reference: `eval_interactive_davis.py:17`, `davis_processor.py:5`; not installed here,
no network).  NOT shipped and NOT used by the product path.

It provides the two entry points the reference touches, with a SCRIPTED robot instead of the framework's error-driven one:
  davisinteractive.session.session.DavisInteractiveSession   next / get_scribbles / submit_masks / get_report / get_global_summary
  davisinteractive.utils.scribbles.scribbles2mask            rasterise scribble paths into a label map
The scribbles depend only on the ground-truth annotations and the interaction counter (never on the submitted masks), so two
implementations of the propagation engine receive the identical interaction sequence and their outputs can be compared."""
