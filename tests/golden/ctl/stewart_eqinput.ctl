      seqfile = ../data/stewart.aa
     treefile = ../data/stewart.trees
      seqtype = 2
        model = 1   * EqualInput (proportional)
    fix_alpha = 1
        alpha = 0
    cleandata = 0
