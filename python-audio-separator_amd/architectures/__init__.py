"""Drop-in architecture plugins: the four classes ``Separator.load_model`` instantiates (separator.py:889-914), same
module and class names as audio_separator/separator/architectures/, every numerical step in libasx.so."""
