"""Import names of the reference package, resolved to THIS repo's plugin classes: lets the reference's own unit tests
(`from audio_separator.separator.common_separator import CommonSeparator`) run unmodified against the drop-in classes
(tests/test_reference_unit_suites.py)."""
