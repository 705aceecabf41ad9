from phamclust_amd.scripts.phamclust import main

if __name__ == "__main__":
    main()
