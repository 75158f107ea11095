"""CPU oracle for the pfann hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  pfann_amd/ never does: the product path fails loudly without its HIP library.

Each module restates, in numpy / torch-CPU / plain C, what one stage of the reference
computes, citing the reference file:line it follows (paths relative to /root/reference):

  segmenter.py   datautil/musicdata.py:21-93       (a1)  pinned by golden vectors
  melspec.py     datautil/melspec.py:19-50          (a2)  PARITY UNPINNED against torchaudio
  encoder.py     model.py:14-153                    (a3-a5) pinned by golden vectors
  search.py      database.py:121 (faiss IndexFlatIP semantics) (a7) definitional
  seqscore.py    database.py:117-166                (a8-a9) pinned by golden vectors
  seqscore_c.c   cpp/seqscore.cpp:23-136            (a9 native) PARITY UNPINNED by execution
                 (the reference file needs faiss headers + libfaiss, absent here, so it is
                 unbuildable in this image; pinned instead by the probed outputs recorded in
                 SURVEY.md §8c and by agreement with seqscore.py where the two paths agree)

Pinning status is detailed in DESIGN.md §Oracle.
"""
