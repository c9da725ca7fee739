"""Result packaging of the evaluator (reference utils/eval.py:5-13)."""
import os
import zipfile


def zip_folder(source_folder, zip_dir):
    """Zips source_folder into the file zip_dir; archive names start at the folder's own name (the layout the DAVIS /
    YouTube-VOS servers expect: Annotations/<sequence>/<frame>.png)."""
    root = os.path.dirname(source_folder)
    with zipfile.ZipFile(zip_dir, 'w', zipfile.ZIP_DEFLATED) as z:
        for folder, _, files in os.walk(source_folder):
            for name in files:
                full = os.path.join(folder, name)
                z.write(full, full[len(root):].strip(os.path.sep))
