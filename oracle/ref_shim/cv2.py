"""Empty stand-in: `/root/reference/inference_core.py:9` imports cv2 but never uses it."""
