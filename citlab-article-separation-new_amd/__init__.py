"""MI355X-native engine for the ARU-Net + GNN hot path of citlab-article-separation-new.

Import it as ``citlab_article_separation_new_amd`` (the alias package at the repo root adds this
directory to its ``__path__``; a directory name with hyphens cannot be imported directly).
"""
